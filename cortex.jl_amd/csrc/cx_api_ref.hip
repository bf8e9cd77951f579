// cx_api_ref.hip — CX_SCHED_REFERENCE: one cx_sweep / cx_sweep_for is ONE update_marginals! of the reference on any graph, loops
// included (/root/reference/src/inference_engine.jl:559-632).  The order of a call is found on a shadow of the readiness state
// (cx_refsched.h), levelled into stages of independent items, and replayed on the device as ONE graph launch; plans are kept per
// (readiness state at the start of the call, request), so the steady state of an iteration — set the priors, call, set the priors,
// call — replays a standing plan and costs the host a hash lookup.

#include <chrono>
#include <mutex>
#include <thread>
#include <cstdio>
#include <memory>

#include "cx_host.h"
#include "cx_refsched.h"

using namespace cxh;
namespace rs = cx::refsched;

namespace {

struct PlanEntry {
    uint64_t key = 0, req_key = 0;
    int32_t *d_rec = nullptr, *d_list = nullptr, *d_wide_rec = nullptr, *d_flat = nullptr;
    double *d_wide_partial = nullptr;           // 64 pairs of scratch per wide item of the widest stage (k_wide_sum)
    int64_t *d_stage_off = nullptr;
    std::vector<int64_t> stage_off, wide_off;
    // chains of pairs that run as scans (cx_refsched.h: ScanStep; cx_planscan.hip): the links' arrays, the steps, the first step of every stage (-1: none)
    std::vector<rs::ScanStep> scans;
    std::vector<int32_t> scan_at;
    int32_t *d_sl_lead_dst = nullptr, *d_sl_lead_var = nullptr, *d_sl_fol_dst = nullptr, *d_sl_prec = nullptr, *d_sl_src_off = nullptr, *d_sl_src = nullptr;
    uint8_t *d_sl_head = nullptr;
    void *d_scan_scratch = nullptr;
    int64_t n_chain_exec = 0;
    // dim 64: a stage is up to four launches of the kernels a dim 64 sweep is made of — sums of a range of a variable's messages into the
    // product table (k_range_sum64), variable→factor sums (k_v2f64), the messages out of observed variables (k_point64), the MFMA rule on
    // a stored variable→factor message (k_rule64w) — over these lists, by stage
    int32_t *d64_prod = nullptr, *d64_v2f_slot = nullptr, *d64_v2f_var = nullptr, *d64_point = nullptr, *d64_rule = nullptr;
    std::vector<int64_t> off64_prod, off64_v2f, off64_point, off64_rule;
    hipGraphExec_t exec = nullptr;
    bool graph_failed = false;
    bool cluster = false;                       // every stage in ONE launch of an XCD-resident cluster (cx_batch.hip: k_ref_cluster)
    std::shared_ptr<const rs::State> post;      // the readiness state the call leaves
    std::vector<int32_t> order;                 // the executions, in the reference's order (cx_ref_trace)
    int64_t n_messages = 0, n_marginals = 0, n_products = 0, rounds = 0, launches = 0, list_entries = 0;
    uint64_t last_used = 0;
    int64_t device_bytes = 0;
};

struct RefSched {
    rs::Wiring W;
    std::shared_ptr<rs::State> S;               // copy-on-write: a cache hit adopts the entry's post state without copying it
    std::shared_ptr<rs::State> spare;           // the private copy the last call displaced: the next writer copies INTO it (at C4 a state is 200 MB — freeing one and
                                                // allocating the next cost 17 ms per call)
    std::vector<PlanEntry> cache;
    std::vector<int32_t> prod_slot;             // segment-tree node -> index in the handle's product store
    std::vector<int32_t> joint_slot;            // factor -> index in the handle's joint store (-1: its joint marginal is not wired)
    // the user's set_value! loop over a list cx_set_messages knows (set_key): the state it leads to, per state it started from — an iteration
    // that re-sets its 2 M priors walks 2 M signals and copies a 200 MB state otherwise, every time, to arrive where it arrived before
    struct SetTrans { uint64_t pre = 0, set_key = 0, used = 0; std::shared_ptr<const rs::State> post; };
    std::vector<SetTrans> set_trans;
    std::vector<int64_t> last_ids;              // the id list of the last long cx_sweep_for, its request and key (a repeated request is compared, not translated again)
    std::vector<int32_t> last_req;
    uint64_t last_key = 0;
    std::vector<int32_t> all_req;               // the request of a plain cx_sweep (every variable that is neither observed nor a stand-in) and its key,
    uint64_t all_key = 0, all_epoch = ~0ull;    // kept while the observed flags stand (2 M ids at C4: 12 ms of host time per call to rebuild and hash)
    int64_t hits = 0, misses = 0;
    uint64_t tick = 0;
    int last = -1;
    bool touched = false;                       // a value was set or a call ran: the wiring can no longer be replaced (cx_graph_wire)
    int64_t max_bytes = (int64_t)4 << 30;       // plans kept: at most max_entries and at most this much device memory (the one in use always stays)
    int max_entries = 16, run_max = 1024;      // C4 as launches, ms per call: every stage a launch 57.6, runs of stages <= 1024 items 52.8, <= 4096 items 102 (one workgroup is slow on a wide stage)
};

RefSched *ref_of(cx_handle *h) { return (RefSched *)h->ref; }

void entry_free(cx_handle *h, PlanEntry &e) {
    if (e.exec) { (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; }
    for (void *p : {(void *)e.d_rec, (void *)e.d_list, (void *)e.d_stage_off, (void *)e.d_wide_rec, (void *)e.d_flat, (void *)e.d_wide_partial, (void *)e.d_sl_lead_dst, (void *)e.d_sl_lead_var,
                    (void *)e.d_sl_fol_dst, (void *)e.d_sl_prec, (void *)e.d_sl_src_off, (void *)e.d_sl_src, (void *)e.d_sl_head, e.d_scan_scratch,
                    (void *)e.d64_prod, (void *)e.d64_v2f_slot, (void *)e.d64_v2f_var, (void *)e.d64_point, (void *)e.d64_rule}) if (p) (void)hipFree(p);
    e.d64_prod = e.d64_v2f_slot = e.d64_v2f_var = e.d64_point = e.d64_rule = nullptr;
    e.d_rec = e.d_list = e.d_wide_rec = e.d_flat = nullptr; e.d_stage_off = nullptr; e.d_wide_partial = nullptr;
    e.d_sl_lead_dst = e.d_sl_lead_var = e.d_sl_fol_dst = e.d_sl_prec = e.d_sl_src_off = e.d_sl_src = nullptr; e.d_sl_head = nullptr; e.d_scan_scratch = nullptr;
    h->device_bytes -= e.device_bytes; e.device_bytes = 0;
}

rs::State &writable(RefSched *R) {
    R->touched = true;
    if (R->S.use_count() > 1) {      // a cached plan still names this state as its result
        std::shared_ptr<rs::State> n = (R->spare && R->spare.use_count() == 1) ? std::move(R->spare) : std::make_shared<rs::State>();
        R->spare.reset();
        *n = *R->S;                  // (vector assignment: the spare's storage is reused)
        R->S = std::move(n);
    }
    return *R->S;
}

// the stages of a plan on the handle's stream: runs of stages of at most run_max items as ONE launch of one workgroup (a barrier
// between the stages instead of a kernel boundary), the wide ones a launch each
static bool flat_runs() { static const bool on = [] { const char *v = std::getenv("CX_REF_FLAT_RUNS"); return !(v && v[0] == '0'); }(); return on; }      // (A/B: 0 = runs on the ordinary records)
int64_t issue(cx_handle *h, RefSched *R, const PlanEntry &e, bool count_only) {
    const size_t ns = e.stage_off.empty() ? 0 : e.stage_off.size() - 1;
    int64_t launches = 0;
    if (!count_only) h->d_ref_list = e.d_list;
    if (cx::is_mfma_dim(h->cfg.dim)) {      // the stage's items sorted into the kernels of a dim 64 sweep (lists made with the plan: plan64)
        for (size_t s = 0; s < ns; s++) {
            const int64_t np = e.off64_prod[s + 1] - e.off64_prod[s], nf = e.off64_v2f[s + 1] - e.off64_v2f[s], npt = e.off64_point[s + 1] - e.off64_point[s], nr = e.off64_rule[s + 1] - e.off64_rule[s];
            if (!count_only) {
                cx::mv64_launch_range_sums(h, (int)np, e.d64_prod + 4 * e.off64_prod[s], h->d_mv_f2v, h->d_mv_prod);
                cx::mv64_launch_v2f(h, (int)nf, e.d64_v2f_slot + e.off64_v2f[s], e.d64_v2f_var + e.off64_v2f[s], h->d_mv_f2v);
                cx::mv64_launch_point(h, (int)npt, e.d64_point + e.off64_point[s], h->d_mv_f2v, h->d_mv_f2v);
                cx::mv64_launch_rule(h, (int)nr, e.d64_rule + 8 * e.off64_rule[s], h->d_mv_f2v, h->d_mv_f2v, CX_KERNEL_BATCH);
            }
            launches += (np > 0) + (nf > 0) + (npt > 0) + (nr > 0);
        }
        return launches;
    }
    if (h->cfg.dim > 1) {      // dim 2 .. 4 (cx_mvbatch.hip): runs of thin stages as one launch of one workgroup, the others a launch of k_batch_mv each; no cluster
        const int64_t thin = std::min<int64_t>(R->run_max, cx::mv_run_block());
        for (size_t s = 0; s < ns;) {
            size_t t = s;
            while (t < ns && e.stage_off[t + 1] - e.stage_off[t] <= thin) t++;
            if (t >= s + 2) { if (!count_only) cx::mv_launch_batch_run(h, e.d_rec, e.d_stage_off, (int)s, (int)t); launches++; s = t; continue; }
            const int64_t n = e.stage_off[s + 1] - e.stage_off[s];
            if (n > 0) { if (!count_only) cx::mv_launch_batch(h, e.d_rec + 5 * e.stage_off[s], n); launches++; }
            s++;
        }
        return launches;
    }
    auto wide_at = [&](size_t s) { return e.wide_off.empty() ? (int64_t)0 : e.wide_off[s + 1] - e.wide_off[s]; };
    // the scan steps of a stage (chains of pairs, cx_planscan.hip): beside the stage's items — they read nothing those write and the other way round
    auto scans_of = [&](size_t s) {
        if (e.scan_at.empty() || e.scan_at[s] < 0) return;
        for (size_t k = (size_t)e.scan_at[s]; k < e.scans.size() && e.scans[k].stage == (int32_t)s + 1; k++) {
            if (!count_only) cx::launch_plan_scan(h, e.d_sl_lead_dst, e.d_sl_lead_var, e.d_sl_fol_dst, e.d_sl_prec, e.d_sl_src_off, e.d_sl_src, e.d_sl_head, e.scans[k].lo, e.scans[k].hi, e.d_scan_scratch);
            launches += 2;
        }
    };
    auto has_scan = [&](size_t s) { return !e.scan_at.empty() && e.scan_at[s] >= 0; };
    for (size_t s = 0; s < ns;) {
        size_t t = s;
        const int64_t thin_max = e.d_flat && flat_runs() ? std::min<int64_t>(R->run_max, cx::flat_run_max()) : R->run_max;
        while (t < ns && e.stage_off[t + 1] - e.stage_off[t] <= thin_max && wide_at(t) == 0 && !has_scan(t)) t++;
        if (t >= s + 2) {
            if (!count_only) { if (e.d_flat && flat_runs()) cx::launch_flat_run(h, e.d_flat, e.d_rec, e.d_stage_off, (int)s, (int)t); else cx::launch_batch_run(h, e.d_rec, e.d_stage_off, (int)s, (int)t); }
            launches++; s = t; continue;
        }
        const int64_t n = e.stage_off[s + 1] - e.stage_off[s], nw = wide_at(s);
        if (n > 0) { if (!count_only) cx::launch_batch(h, e.d_rec + 5 * e.stage_off[s], n); launches++; }
        if (nw > 0) { if (!count_only) cx::launch_wide_sum(h, e.d_wide_rec + 5 * e.wide_off[s], nw, e.d_wide_partial); launches += 2; }      // (independent of the stage's other items)
        scans_of(s);
        s++;
    }
    return launches;
}

int32_t run_entry(cx_handle *h, RefSched *R, PlanEntry &e) {
    if (e.cluster && h->cluster_state > 0 && !h->profiling) {
        h->d_ref_list = e.d_list;
        return cluster_run(h, e.d_flat, e.d_rec, e.d_stage_off, e.stage_off, (int64_t)e.stage_off.size() - 1, &e.launches);
    }
    static const bool graphs = [] { const char *v = std::getenv("CX_REF_GRAPH"); return !(v && v[0] == '0'); }();
    if (graphs && !e.graph_failed && !h->profiling && !e.exec && e.launches > 1) {
        hipError_t er = hipSuccess;
        if (!h->tree_capture_stream) er = hipStreamCreateWithFlags(&h->tree_capture_stream, hipStreamNonBlocking);
        hipGraph_t g = nullptr;
        if (er == hipSuccess) er = hipStreamBeginCapture(h->tree_capture_stream, hipStreamCaptureModeThreadLocal);
        if (er == hipSuccess) {
            hipStream_t user = h->stream;
            h->stream = h->tree_capture_stream;
            (void)issue(h, R, e, false);
            h->stream = user;
            er = hipStreamEndCapture(h->tree_capture_stream, &g);
        }
        if (er == hipSuccess && g) er = hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (er != hipSuccess || !e.exec) { (void)hipGetLastError(); e.exec = nullptr; e.graph_failed = true; }
    }
    if (e.exec && !h->profiling) {
        if (hipGraphLaunch(e.exec, h->stream) == hipSuccess) return CX_OK;
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; e.graph_failed = true;
    }
    (void)issue(h, R, e, false);
    return CX_OK;
}

}  // namespace

namespace cxh {

// The records of a plan with everything the graph's tables would answer already filled in (cx_batch.hip: FlatRec — kind | n << 8,
// destination, variable, five sources): what the XCD-resident cluster runs, so that an item's chain of dependent loads is its values and
// nothing else.  Items that do not fit (more than five sources, rules of factors with more than two edges, variational rules) point back
// at their ordinary record.
void flat_records(const cx_handle *h, const std::vector<int32_t> &rec, const std::vector<int32_t> &list, const std::vector<int64_t> &stage_off, std::vector<int32_t> &flat) {
    constexpr int32_t kSumToFactor = 1, kSumToMarginal = 2, kSumToGamma = 3, kSumToProduct = 4, kRule = 5, kGeneric = 6, kVmp = 7, kCheckObserved = 0x80;
    const int64_t n = (int64_t)rec.size() / 5;
    flat.assign((size_t)8 * n, 0);
    // (a record depends on nothing but itself: large plans are resolved by several threads — 18 M records: 0.2 s on one)
    const int64_t n_threads = n < (1 << 20) ? 1 : std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)std::thread::hardware_concurrency()));
    auto work = [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; i++) {
        const int32_t *r = &rec[5 * i];
        int32_t *o = &flat[8 * i];
        const int32_t pair = r[0] & (rs::kRecLeads | rs::kRecFollows), kind = r[0] & rs::kRecKindMask;      // (the pair flags travel in the flat kind too)
        o[0] = kGeneric | pair; o[1] = (int32_t)i;
        if (kind == CX_ITEM_MESSAGE_TO_FACTOR || kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
            const int32_t v = r[2], deg = h->var_off[v + 1] - h->var_off[v], b = h->vbase[v];
            const int32_t stride = ((h->vinfo[v] & cx::kDegMask) == cx::kBigDeg) ? 1 : cx::kBlock;
            if (kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
                if (deg < 1 || deg > 5) continue;
                o[0] = kSumToMarginal | (deg << 8) | pair; o[1] = v; o[2] = v;
                for (int32_t j = 0; j < deg; j++) o[3 + j] = b + j * stride;
            } else {
                if (deg < 2 || deg > 6) continue;
                const int32_t k = (r[1] - b) / stride;
                int32_t m = 0;
                for (int32_t j = 0; j < deg; j++) if (j != k) o[3 + m++] = b + j * stride;      // ascending: the reference's fold over the other messages
                o[0] = kSumToFactor | kCheckObserved | (m << 8) | pair; o[1] = r[1]; o[2] = v;
            }
        } else if (kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
            const int32_t p = h->partner[r[1]];
            if (p < 0) continue;
            o[0] = kRule | (1 << 8) | pair; o[1] = r[1]; o[2] = r[2]; o[3] = p;
        } else if ((kind == rs::kItemSumToFactor || kind == rs::kItemSumToMarginal || kind == rs::kItemSumToProduct || kind == rs::kItemSumToGammaMarginal) && r[4] >= 1 && r[4] <= 5) {
            o[0] = (kind == rs::kItemSumToFactor ? kSumToFactor : kind == rs::kItemSumToMarginal ? kSumToMarginal : kind == rs::kItemSumToProduct ? kSumToProduct : kSumToGamma) | (r[4] << 8) | pair;
            o[1] = r[1]; o[2] = r[2];
            for (int32_t j = 0; j < r[4]; j++) o[3 + j] = list[r[3] + j];
        } else if (kind >= rs::kItemMfNormal && kind <= rs::kItemStGamma && r[4] >= 1 && r[4] <= 3) {      // a variational rule: which one in the count field
            o[0] = kVmp | ((kind - rs::kItemMfNormal) << 8) | pair; o[1] = r[1]; o[2] = r[2];
            for (int32_t j = 0; j < r[4]; j++) o[3 + j] = list[r[3] + j];
        }
    }
    };
    if (n_threads == 1) work(0, n);
    else {
        std::vector<std::thread> pool;
        for (int64_t t = 0; t < n_threads; t++) pool.emplace_back(work, n * t / n_threads, n * (t + 1) / n_threads);
        for (auto &th : pool) th.join();
    }
    // A stage that a run of one workgroup can take (cx_batch.hip: k_flat_run, at most 960 records): its units — a record, or a leader and its
    // follower — sorted by kind, the first record of every kind flagged (kFlatGroupStart: with at most 64 records each kind gets a wavefront
    // of its own; with more, a wavefront takes a consecutive run of them, mostly of one kind).  The order of the items of a stage is free:
    // they are independent.
    constexpr int32_t kGroupStart = 0x10000000;
    struct Unit { uint64_t key; int32_t first, count; };
    std::vector<Unit> units;
    std::vector<int32_t> tmp;
    for (size_t st = 0; st + 1 < stage_off.size(); st++) {
        const int64_t lo = stage_off[st], W = stage_off[st + 1] - lo;
        if (W < 2 || W > cx::flat_run_max()) continue;      // (group starts matter to the kernel up to 64 records; the order up to a run's widest stage)
        units.clear();
        for (int64_t i = 0; i < W;) {
            const int32_t *o = &flat[8 * (lo + i)];
            const bool leads = (o[0] & rs::kRecLeads) && i + 1 < W;
            auto kind_of = [](int32_t k) { return (uint32_t)((k & 0x7f) == kVmp ? (k & 0xffff) : (k & 0xff)); };      // (a variational rule's name is in the count field)
            const uint64_t k0 = kind_of(o[0]), k1 = leads ? kind_of(flat[8 * (lo + i + 1)]) : 0;
            units.push_back({(k0 << 16) | k1, (int32_t)i, leads ? 2 : 1});
            i += leads ? 2 : 1;
        }
        std::stable_sort(units.begin(), units.end(), [](const Unit &a, const Unit &b) { return a.key < b.key; });
        tmp.assign(flat.begin() + 8 * lo, flat.begin() + 8 * (lo + W));
        int64_t w = 0;
        for (size_t u = 0; u < units.size(); u++) {
            for (int32_t c = 0; c < units[u].count; c++, w++) std::copy(tmp.begin() + 8 * (units[u].first + c), tmp.begin() + 8 * (units[u].first + c) + 8, flat.begin() + 8 * (lo + w));
            if (u == 0 || units[u].key != units[u - 1].key) flat[8 * (lo + w - units[u].count)] |= kGroupStart;
        }
    }
}


bool cluster_prepare(cx_handle *h) {
    if (h->cluster_state != 0) return h->cluster_state > 0;
    h->cluster_state = -1;
    if (const char *v = std::getenv("CX_REF_CLUSTER")) if (v[0] == '0') return false;
    if (const char *v = std::getenv("CX_REF_CLUSTER_MIN")) h->cluster_min_items = std::max<int64_t>(0, std::atoll(v));
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return false; }
    // the cluster rests on gfx942 / gfx950 specifics (XCC_ID in hwreg 20, s_waitcnt vmcnt(0) as a store acknowledgement, one L2 per XCD that serves sc1
    // loads of plain stores): any other device keeps plain launches
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && std::strncmp(prop.gcnArchName, "gfx942", 6) != 0) return false;
    if (hipMalloc(&h->d_cluster_ctl, 512) != hipSuccess) { (void)hipGetLastError(); h->d_cluster_ctl = nullptr; return false; }
    h->cluster_cu = cus;
    h->cluster_state = 1;
    return true;
}

bool cluster_fits(const cx_handle *h, const std::vector<int64_t> &stage_off, int64_t ns) {
    if (h->cluster_state <= 0 || h->cfg.dim != 1 || ns < 8 || (int64_t)stage_off.size() < ns + 1) return false;
    const int64_t items = stage_off[ns] - stage_off[0];
    const int64_t two_gib = (int64_t)1 << 31;
    if (items < h->cluster_min_items * ns || h->nslots * 16 >= two_gib || h->nv * 16 >= two_gib || (int64_t)h->prod_index.size() * 16 >= two_gib) return false;
    // the cluster is an eighth of the chip: it wins on stages it takes in one pass (≈ 3 us against ≈ 6.5 us for a launch) and loses on wider
    // ones, which leave as launches — and every switch between the two is a launch and a wait for the device (≈ 25 us).  Estimated both ways.
    // As launches, consecutive stages of at most 1,024 items fold into one launch of one workgroup (k_batch_run: ≈ 1 us a stage).
    int64_t narrow = 0, wide = 0, segments = 0, thin = 0, thin_runs = 0;
    bool in_seg = false, in_run = false;
    for (int64_t s = 0; s < ns; s++) {
        const int64_t w = stage_off[s + 1] - stage_off[s];
        const bool nar = w <= h->cluster_max_items, th = w <= 1024;
        if (nar) { narrow++; if (!in_seg) segments++; } else wide++;
        if (th) { thin++; if (!in_run) thin_runs++; }
        in_seg = nar; in_run = th;
    }
    const double as_launches = 6.5 * (double)(ns - thin) + 6.5 * (double)thin_runs + 1.0 * (double)thin;
    const double on_cluster = 3.0 * (double)narrow + 6.5 * (double)wide + 25.0 * (double)segments;
    return narrow >= 8 && on_cluster < 0.8 * as_launches;
}

int32_t cluster_run(cx_handle *h, const int32_t *d_flat, const int32_t *d_rec, const int64_t *d_stage_off, const std::vector<int64_t> &stage_off, int64_t ns, int64_t *launches) {
    // stages wider than the whole chip is (the first two of a grid's plan: every prior's message at once) leave as ordinary launches
    // on all eight XCDs; the runs of stages between them go to the cluster, one launch per run
    if (launches) *launches = 0;
    // one cluster at a time per process: two launches that each hold part of one XCD's compute units would wait for each other's workgroups
    // to become resident until both time out (handles driven from different threads; another PROCESS's cluster can still do that: both
    // calls then fail loudly after their bounded waits and their handles go back to launches)
    static std::mutex one_cluster;
    std::lock_guard<std::mutex> hold(one_cluster);
    for (int64_t s = 0; s < ns;) {
        const int64_t w = stage_off[s + 1] - stage_off[s];
        if (launches) ++*launches;
        // (cluster_state < 0: an earlier run of THIS call timed out and was finished on launches — so is the rest of the call)
        if (w > h->cluster_max_items || h->cluster_state <= 0) { if (w > 0) cx::launch_batch(h, d_rec + 5 * stage_off[s], w); s++; continue; }
        int64_t t = s;
        while (t < ns && stage_off[t + 1] - stage_off[t] <= h->cluster_max_items) t++;
        cx::launch_ref_cluster(h, h->d_cluster_ctl, h->cluster_cu, d_flat, d_rec, d_stage_off + s, (int)(t - s));
        // The members' waits are bounded in time.  A run whose cluster gave up has computed its first stages and part of one more; the
        // items of one stage never read each other's outputs (rs::level's invariant: a reader of a value and its next writer are in
        // different stages), so a partly executed stage can be executed again, and a stage is complete exactly when all P members have
        // arrived at its barrier (their stores acknowledged first): the run resumes at stage arrive / P as plain launches — from its
        // first stage when the membership itself was never settled — and the handle stays with launches from then on.
        unsigned ctl[128] = {0};
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipMemcpyAsync(ctl, h->d_cluster_ctl, sizeof(ctl), hipMemcpyDeviceToHost, h->stream));
        CX_HIP(h, hipStreamSynchronize(h->stream));
        if (ctl[4] || ctl[1] == 0) {
            h->cluster_state = -1;
            h->cluster_recoveries++;
            const int64_t P = ctl[7], done = P > 0 ? std::min<int64_t>((int64_t)ctl[3] / P, t - s) : 0;
            h->cluster_note = "a barrier of the XCD-resident cluster timed out (" + std::to_string(ctl[1]) + " member workgroups of " + std::to_string(ctl[0]) + " registered, " +
                              std::to_string(done) + " of " + std::to_string(t - s) + " stages complete); the call was finished on plain launches, as are all further calls of this handle";
            if (std::getenv("CX_REF_CLUSTER_VERBOSE")) std::fprintf(stderr, "[cortex_hip] %s\n", h->cluster_note.c_str());
            for (int64_t x = s + done; x < t; x++) {
                const int64_t wx = stage_off[x + 1] - stage_off[x];
                if (wx > 0) cx::launch_batch(h, d_rec + 5 * stage_off[x], wx);
                if (launches) ++*launches;
            }
            CX_HIP(h, hipGetLastError());
            s = t;
            continue;
        }
        if (std::getenv("CX_REF_CLUSTER_TIME")) {      // member 0's clock (10 ns ticks): issue, memory, workgroup, cluster, release
            unsigned long long tk[5]; std::memcpy(tk, ctl + 64, sizeof(tk));
            std::fprintf(stderr, "[cluster %lld stages] us per stage: issue %.3f memory %.3f workgroup %.3f cluster %.3f release %.3f\n", (long long)(t - s),
                         tk[0] * 0.01 / (t - s), tk[1] * 0.01 / (t - s), tk[2] * 0.01 / (t - s), tk[3] * 0.01 / (t - s), tk[4] * 0.01 / (t - s));
            unsigned long long wk[16]; std::memcpy(wk, ctl + 80, sizeof(wk));
            std::fprintf(stderr, "  member 0's wavefronts, release to acknowledged stores:");
            for (int j = 0; j < 16; j++) std::fprintf(stderr, " %.2f", wk[j] * 0.01 / (t - s));
            std::fprintf(stderr, "\n");
        }
        s = t;
    }
    return CX_OK;
}

void ref_free(cx_handle *h) {
    RefSched *R = ref_of(h);
    if (!R) return;
    for (auto &e : R->cache) entry_free(h, e);
    delete R;
    h->ref = nullptr; h->d_ref_list = nullptr;
}

// the captured graphs hold the product store's address by value: dropped when the store moves
void ref_graphs_drop(cx_handle *h) {
    RefSched *R = ref_of(h);
    if (!R) return;
    for (auto &e : R->cache) if (e.exec) { (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; }
}

// the segment-tree nodes of the variables of degree > 5 live in the handle's product store, under the keys cx_update_batch's
// ProductOfMessages items and cx_get_products use; the wired joint marginals in the joint store, under the keys of cx_get_joint_marginals
static int32_t register_stores(cx_handle *h, RefSched *R) {
    R->prod_slot.resize(R->W.prods.size());
    for (size_t i = 0; i < R->W.prods.size(); i++) {
        const auto &p = R->W.prods[i];
        auto key = std::make_tuple(p.var, p.lo, p.hi);
        auto it = h->prod_index.find(key);
        if (it == h->prod_index.end()) it = h->prod_index.emplace(key, (int32_t)h->prod_index.size()).first;
        R->prod_slot[i] = it->second;
    }
    R->joint_slot.assign(R->W.jrule.size(), -1);
    for (size_t f = 0; f < R->W.jrule.size(); f++) {
        if (!R->W.jrule[f]) continue;
        auto it = h->joint_index.find((int32_t)f);
        if (it == h->joint_index.end()) it = h->joint_index.emplace((int32_t)f, (int32_t)h->joint_index.size()).first;
        R->joint_slot[f] = it->second;
    }
    if (h->cfg.dim > 1) return mv_ensure_prod_store(h);      // (no joint marginals, no variational rules: cx_graph_wire refuses dim > 1)
    { const int32_t rp = ensure_prod_store(h); if (rp != CX_OK) return rp; }
    return ensure_joint_store(h);
}

// the wiring and the shadow state, built with the graph (everything starts as UndefValue(): nothing computed, nothing fresh)
int32_t ref_build(cx_handle *h) {
    if (h->ref) return CX_OK;
    std::unique_ptr<RefSched> R(new RefSched());
    std::string err;
    const int32_t rc = rs::build_wiring(h, R->W, err);
    if (rc != CX_OK) return fail(h, rc, err);
    R->S = std::make_shared<rs::State>();
    rs::init_state(R->W, *R->S);
    { const int32_t rp = register_stores(h, R.get()); if (rp != CX_OK) return rp; }
    if (const char *v = std::getenv("CX_REF_CACHE")) R->max_entries = std::max(1, std::atoi(v));
    (void)cluster_prepare(h);
    if (const char *v = std::getenv("CX_REF_CACHE_MB")) R->max_bytes = std::max<int64_t>(1, std::atoll(v)) << 20;
    if (const char *v = std::getenv("CX_REF_RUN_MAX")) R->run_max = std::min(1024, std::max(0, std::atoi(v)));      // (k_flat_run: a stage is one pass of 1,024 threads)
    h->ref = R.release();
    return CX_OK;
}

// set_value! of message signals through the ABI (cx_set_messages): edges as CSR indices
void ref_on_set(cx_handle *h, int64_t n, const int64_t *edges, int32_t direction, uint64_t set_key) {
    RefSched *R = ref_of(h);
    if (!R || n == 0) return;
    const uint64_t pre = R->S->hash;
    if (set_key)
        for (auto &t : R->set_trans)
            if (t.pre == pre && t.set_key == set_key) {      // the same list from the same state: the same state
                t.used = ++R->tick;
                R->touched = true;
                if (R->S.use_count() == 1 && R->S != t.post) R->spare = std::move(R->S);
                R->S = std::const_pointer_cast<rs::State>(t.post);
                return;
            }
    rs::State &S = writable(R);
    for (int64_t i = 0; i < n; i++) rs::set_value(R->W, S, direction == CX_TO_FACTOR ? R->W.sig_v2f(edges[i]) : R->W.sig_f2v(edges[i]));
    if (set_key) {
        if (R->set_trans.size() >= 3) {
            size_t lru = 0;
            for (size_t j = 1; j < R->set_trans.size(); j++) if (R->set_trans[j].used < R->set_trans[lru].used) lru = j;
            R->set_trans.erase(R->set_trans.begin() + lru);
        }
        R->set_trans.push_back(RefSched::SetTrans{pre, set_key, ++R->tick, R->S});      // (shared from now on: the next writer copies)
    }
}

// set_value! of marginal signals (cx_set_marginals under a user wiring): local variable numbers
void ref_on_set_marginals(cx_handle *h, int64_t n, const int32_t *vars) {
    RefSched *R = ref_of(h);
    if (!R || n == 0) return;
    rs::State &S = writable(R);
    for (int64_t i = 0; i < n; i++) rs::set_value(R->W, S, R->W.sig_marg(vars[i]));
}

// cx_set_marginals on a reference-order handle: the user's set_value!(get_variable_marginal(...), value) — the initial q's and the data of
// a variational wiring (test/inference_engine_tests.jl:717-736).  Store: (mean, variance) of a Normal variable, (shape, scale) of a precision.
int32_t ref_set_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, int32_t form, const double *payload) {
    RefSched *R = ref_of(h);
    CX_REQUIRE(h, R && h->cfg.dim == 1, CX_ERR_STATE, "cx_set_marginals: no reference-order wiring");
    CX_REQUIRE(h, form == CX_FORM_POINT || form == CX_FORM_MOMENT || form == CX_FORM_MEAN_PRECISION || form == CX_FORM_GAMMA, CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: bad form");
    try {
        std::vector<int32_t> vars((size_t)n);
        std::vector<double2> val((size_t)n);
        const int64_t stride = form == CX_FORM_POINT ? 1 : 2;
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(h, variable_ids[i]);
            if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
            const bool gamma = !h->var_gamma.empty() && h->var_gamma[v];
            if (gamma != (form == CX_FORM_GAMMA))
                return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_set_marginals: variable " + std::to_string(variable_ids[i]) + (gamma ? " is a precision (CX_FORM_GAMMA)" : " is a Normal variable (CX_FORM_POINT, CX_FORM_MOMENT or CX_FORM_MEAN_PRECISION)"));
            const double *p = payload + i * stride;
            vars[i] = (int32_t)v;
            val[i] = form == CX_FORM_POINT ? make_double2(p[0], 0.0) : form == CX_FORM_MEAN_PRECISION ? make_double2(p[0], 1.0 / p[1]) : make_double2(p[0], p[1]);
        }
        const int64_t bytes_idx = ((n * 4 + 15) / 16) * 16;
        const int32_t rc = ensure_stage(h, bytes_idx + n * 16);
        if (rc != CX_OK) return rc;
        int32_t *d_idx = (int32_t *)h->d_stage;
        double2 *d_val = (double2 *)((char *)h->d_stage + bytes_idx);
        CX_HIP(h, hipMemcpyAsync(d_idx, vars.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
        CX_HIP(h, hipMemcpyAsync(d_val, val.data(), (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
        cx::launch_scatter(h, h->d_marg, d_idx, d_val, n);
        CX_HIP(h, hipGetLastError());
        CX_HIP(h, hipStreamSynchronize(h->stream));      // the staging vectors die here
        ref_on_set_marginals(h, n, vars.data());
        return CX_OK;
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_set_marginals: host allocation failed"); }
}

// cx_seed_messages: every message of `direction` that a rule computes (its slot has a partner or sits in a factor of more edges) and
// that is still undefined gets a value — the user's set_value! over those signals, in edge order
void ref_on_seed(cx_handle *h, int32_t direction) {
    RefSched *R = ref_of(h);
    if (!R) return;
    rs::State &S = writable(R);
    for (int64_t e = 0; e < h->ne; e++) {
        const int32_t slot = cx::slot_of_edge(h, e);
        // (dim 2 .. 4: the seeding kernel gives a value to the messages of pairwise factors only, cx_mv.hip: k_mv_seed)
        if (h->partner[slot] < 0 && (h->cfg.dim > 1 || h->slot_kary.empty() || h->slot_kary[slot] < 0)) continue;
        const int64_t s = direction == CX_TO_FACTOR ? R->W.sig_v2f(e) : R->W.sig_f2v(e);
        if (!(S.flags[s] & rs::kComputed)) rs::set_value(R->W, S, s);
    }
}

// cx_update_batch: a plug-in's process! of each item's signal (compute! = rule + set_value!, signal.jl:392-410)
void ref_on_batch(cx_handle *h, const cx_item *items, int64_t n) {
    RefSched *R = ref_of(h);
    if (!R) return;
    rs::State &S = writable(R);
    for (int64_t i = 0; i < n; i++) {
        const cx_item &it = items[i];
        int64_t s = -1;
        if (it.kind == CX_ITEM_MESSAGE_TO_FACTOR || it.kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
            const int64_t e = find_edge(h, it.variable_id, it.factor_id);
            if (e >= 0) s = it.kind == CX_ITEM_MESSAGE_TO_FACTOR ? R->W.sig_v2f(e) : R->W.sig_f2v(e);
        } else if (it.kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
            const int64_t v = find_var(h, it.variable_id);
            if (v >= 0) s = R->W.sig_marg(v);
        } else if (it.kind == CX_ITEM_PRODUCT_OF_MESSAGES) {
            const int64_t v = find_var(h, it.variable_id);
            if (v >= 0) s = rs::find_prod(R->W, (int32_t)v, (int32_t)((uint64_t)it.factor_id >> 32), (int32_t)((uint64_t)it.factor_id & 0xffffffffu));
        }
        if (s >= 0) rs::set_value(R->W, S, s);
    }
}

// dim 64: the levelled records of a plan as the lists its stages launch (PlanEntry: d64_*).  What an execution computes does not depend
// on HOW the reference's rule would have folded it: a MessageToFactor is the sum of the variable's other stored messages (k_v2f64; the
// segment-tree nodes a variable of degree > 5 hangs it off hold partial sums of the same messages), a ProductOfMessages node the sum of
// its range (k_range_sum64), a MessageToVariable the rule on the sender's stored variable→factor message — or on its datum —, and a
// marginal is formed from the stored messages when it is read (cx_get_marginals), like under every other dim 64 schedule.
static int32_t plan64(cx_handle *h, RefSched *R, const rs::Plan &P, PlanEntry &e) {
    const size_t ns = P.stage_off.size() - 1;
    std::vector<int32_t> prod, v2f_s, v2f_v, point, rule, slot_var(h->nslots, -1), node_of(h->prod_index.size(), -1);
    for (int64_t ed = 0; ed < h->ne; ed++) slot_var[cx::slot_of_edge(h, ed)] = h->edge_var[ed];
    for (size_t i = 0; i < R->prod_slot.size(); i++) if (R->prod_slot[i] >= 0 && R->prod_slot[i] < (int32_t)node_of.size()) node_of[R->prod_slot[i]] = (int32_t)i;
    e.off64_prod.assign(1, 0); e.off64_v2f.assign(1, 0); e.off64_point.assign(1, 0); e.off64_rule.assign(1, 0);
    for (size_t st = 0; st < ns; st++) {
        for (int64_t i = P.stage_off[st]; i < P.stage_off[st + 1]; i++) {
            const int32_t *r = &P.rec[5 * i];
            const int32_t kind = r[0] & rs::kRecKindMask;
            if (kind == CX_ITEM_MESSAGE_TO_FACTOR || kind == rs::kItemSumToFactor) { v2f_s.push_back(r[1]); v2f_v.push_back(r[2]); }
            else if (kind == CX_ITEM_MESSAGE_TO_VARIABLE) {
                const int32_t dst = r[1], p = h->partner[dst];
                if (p < 0) continue;
                if (h->vinfo[slot_var[p]] & cx::kClamped) {
                    if (h->vinfo[r[2]] & cx::kClamped) return fail(h, CX_ERR_UNSUPPORTED, "reference schedule, dim 64: a message between two observed variables is not computed");
                    point.push_back(p);
                } else rule.insert(rule.end(), {p, -1, -1, -1, h->spdir[p], dst, 1, 0});      // flag 1: the stored variable→factor message of slot p is the input
            } else if (kind == rs::kItemSumToProduct) {
                const int32_t pi = r[1] >= 0 && r[1] < (int32_t)node_of.size() ? node_of[r[1]] : -1;
                if (pi < 0) return fail(h, CX_ERR_STATE, "reference schedule, dim 64: a segment-tree node without a row in the product table");
                const auto &pr = R->W.prods[pi];
                const int32_t stride = cx::slot_stride(h, pr.var);
                prod.insert(prod.end(), {r[1], pr.hi - pr.lo + 1, h->vbase[pr.var] + (pr.lo - 1) * stride, stride});
            } else if (kind == CX_ITEM_INDIVIDUAL_MARGINAL || kind == rs::kItemSumToMarginal) {
                continue;      // formed when read
            } else return fail(h, CX_ERR_UNSUPPORTED, "reference schedule, dim 64: item kind " + std::to_string(kind) + " has no dim 64 kernel (factors of more than two variables are dim 2 .. 4)");
        }
        e.off64_prod.push_back((int64_t)prod.size() / 4); e.off64_v2f.push_back((int64_t)v2f_s.size());
        e.off64_point.push_back((int64_t)point.size()); e.off64_rule.push_back((int64_t)rule.size() / 8);
    }
    int32_t rc;
    if ((rc = dev_upload(h, &e.d64_prod, prod)) != CX_OK || (rc = dev_upload(h, &e.d64_v2f_slot, v2f_s)) != CX_OK || (rc = dev_upload(h, &e.d64_v2f_var, v2f_v)) != CX_OK ||
        (rc = dev_upload(h, &e.d64_point, point)) != CX_OK || (rc = dev_upload(h, &e.d64_rule, rule)) != CX_OK) return rc;
    return CX_OK;
}

static uint64_t request_key(const int32_t *req, int64_t n) {
    uint64_t rk = rs::mix64((uint64_t)n);
    for (int64_t i = 0; i < n; i++) rk = rs::mix64(rk ^ (uint64_t)(uint32_t)req[i]);
    return rk;
}

// ONE update_marginals!(engine, request): req = local variable numbers in request order; key: request_key(req, n) when the caller has it
int32_t ref_sweep(cx_handle *h, const int32_t *req, int64_t n, const uint64_t *key_known) {
    RefSched *R = ref_of(h);
    CX_REQUIRE(h, R, CX_ERR_STATE, "reference schedule: no wiring (cx_graph_create builds it)");
    const uint64_t rk = key_known ? *key_known : request_key(req, n);
    const uint64_t key = R->S->hash;
    int hit = -1;
    for (size_t i = 0; i < R->cache.size(); i++) if (R->cache[i].key == key && R->cache[i].req_key == rk) { hit = (int)i; break; }
    if (hit < 0) {
        try {
            // the scheduler runs on a copy: a call the device cannot replay leaves the shadow where it was
            static const bool timing_b = std::getenv("CX_REF_TIMING") != nullptr;
            const auto b0 = std::chrono::steady_clock::now();
            auto T = std::make_shared<rs::State>(*R->S);
            rs::Call call;
            const int32_t bad = rs::update_marginals(R->W, *T, req, n, call);
            const auto b1 = std::chrono::steady_clock::now();
            (void)bad;      // reported by level() with the message's ids
            rs::Plan P;
            std::string err;
            const int32_t rc = rs::level(h, R->W, call, [&](int64_t i) { return R->prod_slot[i]; }, [&](int64_t f) { return R->joint_slot[f]; }, P, err);
            if (rc != CX_OK) return fail(h, rc, err);
            const auto b2 = std::chrono::steady_clock::now();
            const int64_t incoming = (int64_t)(P.rec.size() + P.list.size() + P.wide_rec.size() + P.rec.size() / 5 * 8) * 4 + (int64_t)P.stage_off.size() * 8;
            auto kept_bytes = [&] { int64_t b = 0; for (auto &c : R->cache) b += c.device_bytes; return b; };
            while (!R->cache.empty() && ((int)R->cache.size() >= R->max_entries || kept_bytes() + incoming > R->max_bytes)) {      // least recently used out
                size_t lru = 0;
                for (size_t i = 1; i < R->cache.size(); i++) if (R->cache[i].last_used < R->cache[lru].last_used) lru = i;
                CX_HIP(h, hipStreamSynchronize(h->stream));
                entry_free(h, R->cache[lru]);
                R->cache.erase(R->cache.begin() + lru);
            }
            PlanEntry e;
            e.key = key; e.req_key = rk;
            e.stage_off = P.stage_off; e.wide_off = P.wide_off;
            e.n_messages = P.n_messages; e.n_marginals = P.n_marginals; e.n_products = P.n_products; e.rounds = P.rounds; e.list_entries = (int64_t)P.list.size();
            // wide and deep: the cluster; chains of thin stages stay with one workgroup's runs (k_batch_run), short plans with plain launches
            e.cluster = P.wide_rec.empty() && P.scans.empty() && cluster_fits(h, P.stage_off, (int64_t)P.stage_off.size() - 1);
            e.scans = P.scans; e.n_chain_exec = P.n_chain_exec;
            if (!P.scans.empty()) {
                e.scan_at.assign(P.stage_off.size() - 1, -1);
                for (size_t k = P.scans.size(); k-- > 0;) e.scan_at[P.scans[k].stage - 1] = (int32_t)k;
            }
            // flat records: the cluster's, and the runs of thin stages' (k_flat_run) — items that load through 2 GiB buffer windows
            const int64_t two_gib = (int64_t)1 << 31;
            const bool want_flat = e.cluster || (h->cfg.dim == 1 && flat_runs() && h->nslots * 16 < two_gib && h->nv * 16 < two_gib && (int64_t)h->prod_index.size() * 16 < two_gib &&
                                                 (int64_t)h->joint_index.size() * 48 < two_gib);
            std::vector<int32_t> flat;
            if (want_flat) flat_records(h, P.rec, P.list, P.stage_off, flat);
            const auto b3 = std::chrono::steady_clock::now();
            const int64_t before = h->device_bytes;
            int32_t rc2;
            if ((rc2 = dev_upload(h, &e.d_rec, P.rec)) != CX_OK || (rc2 = dev_upload(h, &e.d_list, P.list)) != CX_OK || (rc2 = dev_upload(h, &e.d_stage_off, P.stage_off)) != CX_OK ||
                (rc2 = dev_upload(h, &e.d_wide_rec, P.wide_rec)) != CX_OK || (want_flat && (rc2 = dev_upload(h, &e.d_flat, flat)) != CX_OK) ||
                (!P.wide_rec.empty() && (rc2 = dev_alloc(h, &e.d_wide_partial, (int64_t)(P.wide_rec.size() / 5) * 64 * 2)) != CX_OK)) {
                e.device_bytes = h->device_bytes - before; entry_free(h, e); return rc2;
            }
            if (cx::is_mfma_dim(h->cfg.dim) && (rc2 = plan64(h, R, P, e)) != CX_OK) { e.device_bytes = h->device_bytes - before; entry_free(h, e); return rc2; }
            if (!P.scans.empty()) {
                int64_t widest = 0;
                for (auto &sc : P.scans) widest = std::max(widest, sc.hi - sc.lo);
                char *scratch = nullptr;
                if ((rc2 = dev_upload(h, &e.d_sl_lead_dst, P.sl_lead_dst)) != CX_OK || (rc2 = dev_upload(h, &e.d_sl_lead_var, P.sl_lead_var)) != CX_OK ||
                    (rc2 = dev_upload(h, &e.d_sl_fol_dst, P.sl_fol_dst)) != CX_OK || (rc2 = dev_upload(h, &e.d_sl_prec, P.sl_prec)) != CX_OK ||
                    (rc2 = dev_upload(h, &e.d_sl_src_off, P.sl_src_off)) != CX_OK || (rc2 = dev_upload(h, &e.d_sl_src, P.sl_src)) != CX_OK ||
                    (rc2 = dev_upload(h, &e.d_sl_head, P.sl_head)) != CX_OK || (rc2 = dev_alloc(h, &scratch, cx::plan_scan_scratch_bytes(widest))) != CX_OK) {
                    e.d_scan_scratch = scratch; e.device_bytes = h->device_bytes - before; entry_free(h, e); return rc2;
                }
                e.d_scan_scratch = scratch;
            }
            e.device_bytes = h->device_bytes - before;
            CX_HIP(h, hipStreamSynchronize(h->stream));      // the plan's host vectors die here
            if (timing_b) {
                const auto b4 = std::chrono::steady_clock::now();
                auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
                fprintf(stderr, "[ref] new plan (%lld executions): scheduler %.1f ms, levelling %.1f ms, flat records %.1f ms, upload %.1f ms\n", (long long)call.order.size(),
                        ms(b0, b1), ms(b1, b2), ms(b2, b3), ms(b3, b4));
            }
            e.post = T;
            e.order = std::move(call.order);
            e.launches = issue(h, R, e, true);
            if (e.cluster) e.launches = 1;
            R->cache.push_back(std::move(e));
            hit = (int)R->cache.size() - 1;
            R->misses++;
        } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "reference schedule: host allocation failed"); }
    } else R->hits++;
    static const bool timing = std::getenv("CX_REF_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    R->touched = true;
    PlanEntry &e = R->cache[hit];
    e.last_used = ++R->tick;
    R->last = hit;
    { const int32_t rc = cx::kary_upload(h); if (rc != CX_OK) return rc; }
    { const int32_t rc = run_entry(h, R, e); if (rc != CX_OK) return rc; }
    CX_HIP(h, hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    if (R->S.use_count() == 1 && R->S != e.post) R->spare = std::move(R->S);      // nobody else holds the state this call started from: its storage serves the next writer
    R->S = std::const_pointer_cast<rs::State>(e.post);      // shared: the next writer copies
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ref] run %.3f ms, adopt state %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    h->sweeps_done++;
    h->v2f_stale = false;
    if (cx::is_mfma_dim(h->cfg.dim)) { h->point64_dirty = true; h->pot64_fresh = false; }      // (as after a batch: a sweep of another schedule recomputes its constants)
    return CX_OK;
}

// the request of a plain cx_sweep: every variable that is neither observed nor a stand-in, ascending id
int32_t ref_sweep_all(cx_handle *h, int32_t n_sweeps) {
    RefSched *R = ref_of(h);
    CX_REQUIRE(h, R, CX_ERR_STATE, "reference schedule: no wiring (cx_graph_create builds it)");
    if (R->all_epoch != h->vinfo_epoch) {
        R->all_req.clear();
        for (int64_t v = 0; v < h->nv; v++) if (!(h->vinfo[v] & (cx::kClamped | cx::kGhost))) R->all_req.push_back((int32_t)v);
        R->all_key = request_key(R->all_req.data(), (int64_t)R->all_req.size());
        R->all_epoch = h->vinfo_epoch;
    }
    for (int32_t s = 0; s < n_sweeps; s++) { const int32_t rc = ref_sweep(h, R->all_req.data(), (int64_t)R->all_req.size(), &R->all_key); if (rc != CX_OK) return rc; }
    return CX_OK;
}

// checkpoint: the shadow as one host section (hash | flags | chunks)
int64_t ref_state_bytes(cx_handle *h) {
    RefSched *R = ref_of(h);
    return R ? 8 + (int64_t)R->S->flags.size() + 8 * (int64_t)R->S->chunks.size() : 0;
}
void ref_state_write(cx_handle *h, char *out) {
    RefSched *R = ref_of(h);
    if (!R) return;
    std::memcpy(out, &R->S->hash, 8);
    std::memcpy(out + 8, R->S->flags.data(), R->S->flags.size());
    std::memcpy(out + 8 + R->S->flags.size(), R->S->chunks.data(), 8 * R->S->chunks.size());
}
bool ref_state_read(cx_handle *h, const char *in, int64_t bytes) {
    RefSched *R = ref_of(h);
    if (!R || bytes != ref_state_bytes(h)) return false;
    auto T = std::make_shared<rs::State>();
    rs::init_state(R->W, *T);            // the static Intermediate / Weak bits are the WIRING's (add_dependency! kwargs), never the blob's
    std::memcpy(T->flags.data(), in + 8, T->flags.size());
    const char *cw = in + 8 + T->flags.size();
    constexpr uint64_t kDynamic = rs::kAllComp | rs::kAllFresh, kStatic = ~kDynamic;
    for (size_t c = 0; c < T->chunks.size(); c++) {
        uint64_t w;
        std::memcpy(&w, cw + 8 * c, 8);
        // a blob of the right size written under ANOTHER wiring would change which dependencies count as weak: refuse it
        if ((w & kStatic) != (T->chunks[c] & kStatic)) return false;
        T->chunks[c] |= w & kDynamic;
    }
    R->touched = true;                   // the imported readiness is state: a later cx_graph_wire must not silently throw it away
    // the fingerprint is recomputed, not trusted
    T->hash = 0;
    for (int64_t c = 0; c < (int64_t)T->chunks.size(); c++) T->hash ^= rs::zob_chunk(c, T->chunks[c]);
    for (int64_t s = 0; s < (int64_t)T->flags.size(); s++) T->hash ^= rs::zob_flag(s, T->flags[s]);
    R->S = T;
    return true;
}

}  // namespace cxh

extern "C" {

int32_t cx_sweep_for(cx_handle *h, int64_t n, const int64_t *variable_ids) {
    CX_NOT_VMP(h, "cx_sweep_for");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_sweep_for: no graph");
    CX_REQUIRE(h, h->cfg.schedule == CX_SCHED_REFERENCE, CX_ERR_UNSUPPORTED,
               "cx_sweep_for: a request for some variables, in the caller's order, is served by CX_SCHED_REFERENCE (on a forest it computes exactly the "
               "messages those marginals need, level by level); the other schedules compute every message: cx_sweep");
    CX_REQUIRE(h, n >= 0 && (n == 0 || variable_ids), CX_ERR_INVALID_ARGUMENT, "cx_sweep_for: null argument");
    if (h->cfg.dim > 1) { const int32_t rp = mv_check_psets(h); if (rp != CX_OK) return rp; }
    try {
        RefSched *R = ref_of(h);
        if (R && n >= 4096 && (int64_t)R->last_ids.size() == n && std::memcmp(R->last_ids.data(), variable_ids, (size_t)n * 8) == 0)
            return ref_sweep(h, R->last_req.data(), n, &R->last_key);
        std::vector<int32_t> req((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t v = find_var(h, variable_ids[i]);
            if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "unknown variable id " + std::to_string(variable_ids[i]));
            req[i] = (int32_t)v;
        }
        if (R && n >= 4096) {
            R->last_ids.assign(variable_ids, variable_ids + n);
            R->last_key = request_key(req.data(), n);
            R->last_req = req;
            return ref_sweep(h, R->last_req.data(), n, &R->last_key);
        }
        return ref_sweep(h, req.data(), n, nullptr);
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_sweep_for: host allocation failed"); }
}

// A user resolver's wiring in place of the default one (src/dependencies.jl:1-15; add_dependency!, src/signal.jl:286-337): the i-th triple
// is add_dependency!(signals[i], dependencies[i]; weak, intermediate, listen) — see cx_refsched.h: build_user_wiring for what may depend
// on what.  Replaces the whole wiring; before the first value is set.
int32_t cx_graph_wire(cx_handle *h, int64_t n, const cx_item *signals, const cx_item *dependencies, const int32_t *flags) {
    CX_NOT_VMP(h, "cx_graph_wire");
    CX_REQUIRE(h, h && h->has_graph, CX_ERR_STATE, "cx_graph_wire: no graph");
    CX_REQUIRE(h, h->cfg.schedule == CX_SCHED_REFERENCE, CX_ERR_UNSUPPORTED, "cx_graph_wire: a dependency wiring is what CX_SCHED_REFERENCE runs the reference's scheduler on; the other schedules are fixed orders");
    // (round 6) dim 2 .. 4 too: a MessageToFactor / marginal under a user wiring is the sum of ITS dependency list, an item k_batch_mv has
    // (the segment-tree signals use it); dim 64 forms marginals from ALL stored messages when they are read, which a wiring may not list
    CX_REQUIRE(h, h->cfg.dim >= 1 && h->cfg.dim <= 4, CX_ERR_UNSUPPORTED, "cx_graph_wire: user wirings run as sums of dependency lists: dim 1 (all rules) and dim 2, 3, 4 (sum-product rules)");
    CX_REQUIRE(h, n >= 0 && (n == 0 || (signals && dependencies && flags)), CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: null argument");
    RefSched *R = ref_of(h);
    CX_REQUIRE(h, R && !R->touched, CX_ERR_STATE, "cx_graph_wire: the wiring is fixed once a value has been set or a call has run (wire right after cx_graph_create, as the reference wires at engine construction)");
    try {
        auto number = [&](const cx_item &it, int64_t *out) -> int32_t {
            if (it.kind == CX_ITEM_INDIVIDUAL_MARGINAL) {
                const int64_t v = find_var(h, it.variable_id);
                if (v < 0) return fail(h, CX_ERR_NOT_FOUND, "cx_graph_wire: unknown variable id " + std::to_string(it.variable_id));
                *out = R->W.sig_marg(v);
                return CX_OK;
            }
            if (it.kind == CX_ITEM_JOINT_MARGINAL) {
                auto ft = std::lower_bound(h->fac_ids.begin(), h->fac_ids.end(), it.factor_id);
                if (ft == h->fac_ids.end() || *ft != it.factor_id) return fail(h, CX_ERR_NOT_FOUND, "cx_graph_wire: unknown factor id " + std::to_string(it.factor_id));
                *out = 2 * h->ne + h->nv + (ft - h->fac_ids.begin());      // (Wiring::sig_joint of the wiring being built)
                return CX_OK;
            }
            if (it.kind != CX_ITEM_MESSAGE_TO_FACTOR && it.kind != CX_ITEM_MESSAGE_TO_VARIABLE)
                return fail(h, CX_ERR_UNSUPPORTED, "cx_graph_wire: signals are MessageToFactor, MessageToVariable, IndividualMarginal and JointMarginal (kind " + std::to_string(it.kind) + ")");
            const int64_t e = find_edge(h, it.variable_id, it.factor_id);
            if (e < 0) return fail(h, CX_ERR_NOT_FOUND, "cx_graph_wire: no connection between variable " + std::to_string(it.variable_id) + " and factor " + std::to_string(it.factor_id));
            *out = it.kind == CX_ITEM_MESSAGE_TO_FACTOR ? R->W.sig_v2f(e) : R->W.sig_f2v(e);
            return CX_OK;
        };
        std::vector<int64_t> s((size_t)n), d((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            if (flags[i] & ~31) return fail(h, CX_ERR_INVALID_ARGUMENT, "cx_graph_wire: flags are CX_WIRE_WEAK | CX_WIRE_INTERMEDIATE | CX_WIRE_NO_LISTEN, or CX_WIRE_DEFAULT_VARIABLE, or CX_WIRE_LINK");
            int32_t rc = number(signals[i], &s[i]);
            d[i] = s[i];
            if (rc == CX_OK && !(flags[i] & CX_WIRE_DEFAULT_VARIABLE)) rc = number(dependencies[i], &d[i]);
            if (rc != CX_OK) return rc;
        }
        rs::Wiring W;
        std::string err;
        const int32_t rc = rs::build_user_wiring(h, n, s.data(), d.data(), flags, W, err);
        if (rc != CX_OK) return fail(h, rc, err);
        CX_HIP(h, hipStreamSynchronize(h->stream));
        for (auto &e : R->cache) entry_free(h, e);
        R->cache.clear(); R->last = -1; R->set_trans.clear();
        R->W = std::move(W);
        R->S = std::make_shared<rs::State>();
        rs::init_state(R->W, *R->S);
        return register_stores(h, R);
    } catch (const std::bad_alloc &) { return fail(h, CX_ERR_OUT_OF_MEMORY, "cx_graph_wire: host allocation failed"); }
}

int32_t cx_ref_plan_stats(const cx_handle *hc, int64_t *out8) {
    cx_handle *h = const_cast<cx_handle *>(hc);
    CX_REQUIRE(h, h && out8, CX_ERR_INVALID_ARGUMENT, "cx_ref_plan_stats: null argument");
    for (int i = 0; i < 8; i++) out8[i] = 0;
    RefSched *R = ref_of(h);
    if (!R || R->last < 0 || R->last >= (int)R->cache.size()) return CX_OK;
    const PlanEntry &e = R->cache[R->last];
    out8[0] = e.stage_off.empty() ? 0 : (int64_t)e.stage_off.size() - 1; out8[1] = e.launches; out8[2] = (int64_t)e.order.size(); out8[3] = e.n_messages;
    out8[4] = e.rounds; out8[5] = (int64_t)R->cache.size(); out8[6] = R->hits; out8[7] = R->misses;
    return CX_OK;
}

int32_t cx_cluster_stats(const cx_handle *hc, int64_t *out4) {
    cx_handle *h = const_cast<cx_handle *>(hc);
    CX_REQUIRE(h, h && out4, CX_ERR_INVALID_ARGUMENT, "cx_cluster_stats: null argument");
    RefSched *R = ref_of(h);
    out4[0] = h->cluster_state; out4[1] = h->cluster_cu; out4[2] = h->cluster_recoveries;
    out4[3] = (R && R->last >= 0 && R->last < (int)R->cache.size() && R->cache[R->last].cluster && h->cluster_state > 0) ? 1 : 0;
    return CX_OK;
}

int32_t cx_ref_trace(const cx_handle *hc, int64_t capacity, cx_item *out, int64_t *n_executions) {
    cx_handle *h = const_cast<cx_handle *>(hc);
    CX_REQUIRE(h, h && n_executions, CX_ERR_INVALID_ARGUMENT, "cx_ref_trace: null argument");
    *n_executions = 0;
    RefSched *R = ref_of(h);
    if (!R || R->last < 0 || R->last >= (int)R->cache.size()) return CX_OK;
    const PlanEntry &e = R->cache[R->last];
    *n_executions = (int64_t)e.order.size();
    if (!out) return CX_OK;
    const int64_t ne = R->W.ne, nv = R->W.nv;
    for (int64_t i = 0; i < std::min<int64_t>(capacity, (int64_t)e.order.size()); i++) {
        const int64_t s = e.order[i];
        cx_item it{};
        if (s < 2 * ne) {
            const int64_t ed = s < ne ? s : s - ne;
            it.kind = s < ne ? CX_ITEM_MESSAGE_TO_FACTOR : CX_ITEM_MESSAGE_TO_VARIABLE;
            it.variable_id = h->var_ids[h->edge_var[ed]]; it.factor_id = h->edge_fac_id[ed];
        } else if (s < 2 * ne + nv) {
            it.kind = CX_ITEM_INDIVIDUAL_MARGINAL; it.variable_id = h->var_ids[s - 2 * ne];
        } else if (R->W.is_joint(s)) {
            it.kind = CX_ITEM_JOINT_MARGINAL; it.factor_id = h->fac_ids[s - R->W.sig_joint(0)];
        } else {
            const auto &p = R->W.prods[s - R->W.sig_prod(0)];
            it.kind = CX_ITEM_PRODUCT_OF_MESSAGES; it.variable_id = h->var_ids[p.var]; it.factor_id = CX_ITEM_RANGE(p.lo, p.hi);
        }
        out[i] = it;
    }
    return CX_OK;
}

}  // extern "C"
