/*
 * oracle/cortex_ref.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the Cortex.jl CPU path that the MI355X sweep replaces:
 * the reactive Signal runtime, the default belief-propagation dependency
 * resolver, the InferenceEngine scheduler and the processor rules that the
 * reference only holds in its test-suite.  Citations are to /root/reference
 * (Cortex.jl v0.3.0); nothing here is copied, every function restates the
 * behaviour of the cited lines.
 *
 * Pinning (see tests/test_oracle_reference_kats.py): schedule, readiness bits,
 * wiring and the conjugate Beta-Bernoulli answer are pinned by the reference's
 * own known-answer tests (test/signal_tests.jl, test/dependencies_tests.jl,
 * test/inference_engine_tests.jl:93-239,360-376,1226-1261).  The reference has
 * NO numeric golden vector for Gaussian messages (only the inequalities at
 * test/inference_engine_tests.jl:485-487, which are checked too), and Julia is
 * not installed here, so Gaussian numerics are pinned by exact tridiagonal /
 * dense solves (oracle/exact.py).  Neighbour iteration order comes from the
 * un-vendored BipartiteFactorGraphs.jl (Project.toml:17); this restatement
 * fixes ascending id — "neighbour order: parity unpinned" beyond the tracing
 * test's [MsgToVar(p,f1), MsgToVar(p,f2)] order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ---------------------------------------------------------------- values -- */

enum { CXO_UNDEF = 0, CXO_REAL = 1, CXO_NORMAL = 2, CXO_BETA = 3, CXO_BOOL = 4,
       /* value kinds only the callback processor produces (test/runtests.jl:48-76): */
       CXO_NORMAL_MP = 5,   /* NormalMeanPrecision: a = mean, b = precision */
       CXO_GAMMA = 6,       /* Gamma: a = shape, b = scale */
       CXO_MVNORMAL2 = 7 }; /* MvNormalMeanPrecision, 2-D: a, b = mean; x = precision row-major */

typedef struct {
    int32_t tag;   /* CXO_UNDEF ≙ UndefValue() (signal.jl:7) */
    double a;      /* REAL: value; NORMAL: mean; BETA: a; BOOL: 0/1 */
    double b;      /* NORMAL: variance; BETA: b */
    double x[4];   /* extra payload of the wider value kinds */
} cxo_value;

/* InferenceSignalVariants (inference_signal.jl:8-97) */
enum {
    CXO_VAR_UNSPECIFIED = 0,
    CXO_VAR_MSG_TO_FACTOR = 1,
    CXO_VAR_MSG_TO_VARIABLE = 2,
    CXO_VAR_PRODUCT = 3,
    CXO_VAR_MARGINAL = 4,
    CXO_VAR_JOINT = 5
};

/* nibble layout, signal.jl:507-519 */
#define NIB_INTERMEDIATE 0x1ull
#define NIB_WEAK 0x2ull
#define NIB_COMPUTED 0x4ull
#define NIB_FRESH 0x8ull
#define ALL_WEAK 0x2222222222222222ull
#define ALL_COMPUTED 0x4444444444444444ull
#define ALL_FRESH 0x8888888888888888ull
#define ALL_PASS 0x1111111111111111ull

typedef struct {
    cxo_value value;
    int32_t variant;
    int64_t variable_id, factor_id; /* tags of the variant */
    int32_t range_lo, range_hi;     /* ProductOfMessages range (1-based, inclusive) */
    uint8_t potentially_pending, pending; /* SignalProps, signal.jl:48-51 */
    int32_t ndeps, capdeps;
    int32_t *deps;
    int32_t nchunks;
    uint64_t chunk0;   /* SignalDependenciesProps always owns >=1 chunk (signal.jl:40-44) */
    uint64_t *xchunks; /* chunks 1.. (heap) */
    int32_t nlist, caplist;
    int32_t *listeners;
    uint8_t *listenmask;
} cxo_signal;

/* factor "functional forms" understood by the restated test processors */
enum {
    CXO_F_OPAQUE = 0,     /* never computed through (e.g. :prior in tests) */
    CXO_F_GAUSS_ADD = 1,  /* likelihood/transition of test/inference_engine_tests.jl:415-432, variance p0 */
    CXO_F_BERNOULLI = 2,  /* test/inference_engine_tests.jl:247-262 */
    CXO_F_DOUBLE = 3      /* :likelihood1/2 of the tracing test (returns 2*dep) :1155-1171 */
};

enum { CXO_P_SSM_BP = 0, CXO_P_BETA_BERNOULLI = 1, CXO_P_TRACING = 2,
       CXO_P_CALLBACK = 3 }; /* rules supplied by the test as a callback (the reference's processors are user code) */

/* rule callback: fills out7 = {a, b, x0..x3} and *tag for signal `sig`; returns 1 on success, 0 for "no rule" */
typedef int32_t (*cxo_rule_cb)(void *ctx, int32_t sig, int32_t *tag, double *out6);

typedef struct {
    int32_t kind; /* 0 none, 1 variable, 2 factor */
    int32_t fkind;
    double p0, p1;
    int32_t marginal; /* signal id (variables) */
    int32_t nlinked, caplinked;
    int32_t *linked;  /* Variable.linked_signals (model_engine.jl:30-35,75-83) */
    int32_t nnb, capnb;
    int64_t *nb;      /* neighbour ids, ascending after finalize */
    int32_t *nbedge;  /* edge index per neighbour */
} cxo_node;

typedef struct {
    int64_t var, fac;
    int32_t m2v, m2f; /* Connection.message_to_variable / message_to_factor (model_engine.jl:181-186) */
} cxo_edge;

typedef struct {
    int64_t round;
    int64_t variable_id;
    int32_t signal;
    cxo_value before, after;
} cxo_exec;

typedef struct {
    cxo_node *nodes; /* index = id (1-based) */
    int64_t nnodes, capnodes;
    cxo_edge *edges;
    int64_t nedges, capedges;
    cxo_signal *sig;
    int64_t nsig, capsig;
    int64_t *var_ids, *fac_ids; /* ascending, filled at finalize */
    int64_t nvars, nfacs;
    int32_t nwarnings;
    int64_t *warn_ctx;
    int32_t processor;
    /* trace */
    int32_t trace_on;
    cxo_exec *trace;
    int64_t ntrace, captrace;
    int64_t nrounds; /* non-empty rounds recorded by last update (inference_engine.jl:818) */
    int64_t cur_round_execs;
    int64_t n_msg_updates, n_marginal_updates; /* counters for the CPU baseline */
    int32_t error; /* sticky error code: 1 = compute on non-pending (signal.jl:399-405), 2 = rule error */
    /* scanner */
    int32_t *scan_out;
    int64_t nscan, capscan;
    int32_t scanning;
    cxo_rule_cb rule_cb;
    void *rule_ctx;
} cxo_engine;

/* ---------------------------------------------------------------- signals -- */

static int32_t sig_new(cxo_engine *E) {
    if (E->nsig == E->capsig) {
        E->capsig = E->capsig ? E->capsig * 2 : 64;
        E->sig = (cxo_signal *)realloc(E->sig, (size_t)E->capsig * sizeof(cxo_signal));
    }
    cxo_signal *s = &E->sig[E->nsig];
    memset(s, 0, sizeof(*s));
    s->nchunks = 1;
    return (int32_t)E->nsig++;
}

static inline uint64_t *chunk_ptr(cxo_signal *s, int32_t c) { return c == 0 ? &s->chunk0 : &s->xchunks[c - 1]; }

/* signal.jl:522-526 (index is 1-based there; 0-based here) */
static inline void nib_set(cxo_signal *s, int32_t i, uint64_t m) { *chunk_ptr(s, i >> 4) |= m << ((i & 15) << 2); }
static inline int nib_get(cxo_signal *s, int32_t i, uint64_t m) { return (*chunk_ptr(s, i >> 4) & (m << ((i & 15) << 2))) != 0; }

/* signal.jl:529-544 */
static int32_t props_add(cxo_signal *s) {
    int32_t newlen = ++s->ndeps;
    int32_t need = (4 * newlen - 1) / 64 + 1;
    if (s->nchunks < need) {
        s->xchunks = (uint64_t *)realloc(s->xchunks, (size_t)(need - 1) * sizeof(uint64_t));
        s->xchunks[need - 2] = 0;
        s->nchunks = need;
    }
    return newlen - 1;
}

/* signal.jl:668-730 — C & (W | F) for every dependency, chunk-parallel */
static int meets_pending_criteria(cxo_signal *s) {
    int32_t n = s->ndeps;
    if (n == 0) return 0;
    for (int32_t c = 0; c < s->nchunks - 1; c++) {
        uint64_t ch = *chunk_ptr(s, c);
        uint64_t W = (ch & ALL_WEAK) >> 1, C = (ch & ALL_COMPUTED) >> 2, F = (ch & ALL_FRESH) >> 3;
        if ((C & (W | F)) != ALL_PASS) return 0;
    }
    int32_t last = (n - 1) >> 4;
    int32_t off = ((n - 1) & 15) << 2;
    /* Julia's `<<` by 64 yields 0; C's is undefined, so branch (signal.jl:716) */
    uint64_t fill = (off + 4 >= 64) ? 0ull : (0xffffffffffffffffull << (off + 4));
    uint64_t ch = *chunk_ptr(s, last) | fill;
    uint64_t W = (ch & ALL_WEAK) >> 1, C = (ch & ALL_COMPUTED) >> 2, F = (ch & ALL_FRESH) >> 3;
    return (C & (W | F)) == ALL_PASS;
}

/* signal.jl:141-154 */
static int sig_is_pending(cxo_engine *E, int32_t id) {
    cxo_signal *s = &E->sig[id];
    if (s->pending) return 1;
    if (s->potentially_pending) {
        int p = meets_pending_criteria(s);
        s->potentially_pending = 0;
        s->pending = (uint8_t)p;
        return p;
    }
    return 0;
}

static inline int sig_is_computed(cxo_engine *E, int32_t id) { return E->sig[id].value.tag != CXO_UNDEF; }

/* signal.jl:339-356 */
static void notify_listener(cxo_engine *E, int32_t listener, int32_t signal, int update_pp) {
    cxo_signal *L = &E->sig[listener];
    if (update_pp) { L->potentially_pending = 1; L->pending = 0; }
    for (int32_t i = 0; i < L->ndeps; i++) {
        if (L->deps[i] == signal) { /* first match only: duplicates never notified (:344) */
            nib_set(L, i, NIB_FRESH);
            nib_set(L, i, NIB_COMPUTED);
            break;
        }
    }
}

/* signal.jl:232-253 */
static void sig_set_value(cxo_engine *E, int32_t id, cxo_value v) {
    cxo_signal *s = &E->sig[id];
    s->value = v;
    for (int32_t c = 0; c < s->nchunks; c++) *chunk_ptr(s, c) &= ~ALL_FRESH; /* :241 */
    s->potentially_pending = 0; s->pending = 0;
    for (int32_t k = 0; k < s->nlist; k++) notify_listener(E, s->listeners[k], id, s->listenmask[k]);
}

/* signal.jl:286-337 */
static void sig_add_dependency(cxo_engine *E, int32_t id, int32_t dep, int weak, int listen, int check_computed, int intermediate) {
    if (id == dep) return; /* :295 */
    cxo_signal *s = &E->sig[id];
    int32_t idx = props_add(s);
    if (weak) nib_set(s, idx, NIB_WEAK);
    if (intermediate) nib_set(s, idx, NIB_INTERMEDIATE);
    if (s->ndeps > s->capdeps) {
        s->capdeps = s->capdeps ? s->capdeps * 2 : 4;
        s->deps = (int32_t *)realloc(s->deps, (size_t)s->capdeps * sizeof(int32_t));
    }
    s->deps[idx] = dep;
    cxo_signal *d = &E->sig[dep];
    if (d->nlist == d->caplist) {
        d->caplist = d->caplist ? d->caplist * 2 : 4;
        d->listeners = (int32_t *)realloc(d->listeners, (size_t)d->caplist * sizeof(int32_t));
        d->listenmask = (uint8_t *)realloc(d->listenmask, (size_t)d->caplist);
    }
    d->listenmask[d->nlist] = (uint8_t)(listen ? 1 : 0);
    d->listeners[d->nlist++] = id;
    s = &E->sig[id];
    if (check_computed && sig_is_computed(E, dep)) {
        nib_set(s, idx, NIB_COMPUTED);
        if (!sig_is_computed(E, id)) nib_set(s, idx, NIB_FRESH);
        s->potentially_pending = 1; s->pending = 0;
    } else if (check_computed) {
        s->potentially_pending = 0; s->pending = 0;
    }
}

/* ------------------------------------------------------------ rule sets -- */

/* product(::NormalMeanVariance, ::NormalMeanVariance), test/runtests.jl:40-46 — same op order */
static cxo_value normal_product(cxo_value l, cxo_value r) {
    double xi = l.a / l.b + r.a / r.b;
    double w = 1 / l.b + 1 / r.b;
    double variance = 1 / w;
    double mean = variance * xi;
    cxo_value o = { CXO_NORMAL, mean, variance, { 0, 0, 0, 0 } };
    return o;
}

/* reduce(product, get_value.(deps)) — left fold in dependency order
 * (test/inference_engine_tests.jl:392,402,412) */
static int fold_normal(cxo_engine *E, cxo_signal *s, cxo_value *out) {
    if (s->ndeps < 1) return 0;
    cxo_value acc = E->sig[s->deps[0]].value;
    if (acc.tag != CXO_NORMAL) return 0;
    for (int32_t i = 1; i < s->ndeps; i++) {
        cxo_value n = E->sig[s->deps[i]].value;
        if (n.tag != CXO_NORMAL) return 0;
        acc = normal_product(acc, n);
    }
    *out = acc;
    return 1;
}

/* test/inference_engine_tests.jl:274-294: Beta(a+a'-1, b+b'-1) fold */
static int fold_beta(cxo_engine *E, cxo_signal *s, cxo_value *out) {
    if (s->ndeps < 1) return 0;
    cxo_value acc = E->sig[s->deps[0]].value;
    if (acc.tag != CXO_BETA) return 0;
    for (int32_t i = 1; i < s->ndeps; i++) {
        cxo_value n = E->sig[s->deps[i]].value;
        if (n.tag != CXO_BETA) return 0;
        acc.a = acc.a + n.a - 1;
        acc.b = acc.b + n.b - 1;
    }
    *out = acc;
    return 1;
}

/* the `variant isa` chain of process! (inference_engine.jl:479-509) dispatching into the
 * restated test processors */
static int apply_rule(cxo_engine *E, int32_t id, cxo_value *out) {
    cxo_signal *s = &E->sig[id];
    switch (E->processor) {
    case CXO_P_CALLBACK: {
        if (!E->rule_cb) return 0;
        double six[6] = { 0, 0, 0, 0, 0, 0 };
        int32_t tag = CXO_UNDEF;
        if (!E->rule_cb(E->rule_ctx, id, &tag, six) || tag == CXO_UNDEF) return 0;
        out->tag = tag; out->a = six[0]; out->b = six[1];
        for (int k = 0; k < 4; k++) out->x[k] = six[2 + k];
        return 1;
    }
    case CXO_P_SSM_BP: /* test/inference_engine_tests.jl:383-432 */
        if (s->variant == CXO_VAR_MSG_TO_VARIABLE) {
            if (s->ndeps != 1) return 0; /* @assert length(dependencies) == 1 */
            cxo_node *f = &E->nodes[s->factor_id];
            if (f->fkind != CXO_F_GAUSS_ADD) return 0;
            cxo_value in = E->sig[s->deps[0]].value;
            if (in.tag == CXO_REAL) { out->tag = CXO_NORMAL; out->a = in.a; out->b = f->p0; return 1; }
            if (in.tag == CXO_NORMAL) { out->tag = CXO_NORMAL; out->a = in.a; out->b = in.b + f->p0; return 1; }
            return 0;
        }
        if (s->variant == CXO_VAR_MSG_TO_FACTOR || s->variant == CXO_VAR_MARGINAL || s->variant == CXO_VAR_PRODUCT)
            return fold_normal(E, s, out);
        return 0;
    case CXO_P_BETA_BERNOULLI: /* test/inference_engine_tests.jl:245-316 */
        if (s->variant == CXO_VAR_MSG_TO_VARIABLE) {
            cxo_node *f = &E->nodes[s->factor_id];
            if (f->fkind != CXO_F_BERNOULLI) return 0;
            cxo_value in = E->sig[s->deps[0]].value;
            if (in.tag != CXO_BOOL) return 0;
            double r = in.a;
            out->tag = CXO_BETA; out->a = 1.0 + r; out->b = 2.0 - r;
            return 1;
        }
        if (s->variant == CXO_VAR_MARGINAL || s->variant == CXO_VAR_PRODUCT) return fold_beta(E, s, out);
        return 0; /* "Should not be invoked" */
    case CXO_P_TRACING: /* test/inference_engine_tests.jl:1153-1181 */
        if (s->variant == CXO_VAR_MSG_TO_VARIABLE) {
            cxo_node *f = &E->nodes[s->factor_id];
            if (f->fkind != CXO_F_DOUBLE) return 0;
            cxo_value in = E->sig[s->deps[0]].value;
            if (in.tag != CXO_REAL) return 0;
            out->tag = CXO_REAL; out->a = 2 * in.a; out->b = 0;
            return 1;
        }
        if (s->variant == CXO_VAR_MARGINAL) {
            double acc = 0;
            for (int32_t i = 0; i < s->ndeps; i++) acc += E->sig[s->deps[i]].value.a;
            out->tag = CXO_REAL; out->a = acc; out->b = 0;
            return 1;
        }
        return 0;
    }
    return 0;
}

/* compute! (signal.jl:392-410) + process! (inference_engine.jl:479-509) */
static void process(cxo_engine *E, int64_t variable_id, int32_t id) {
    if (E->scanning) { /* InferenceRequestScanner, inference_engine.jl:528-537 */
        if (E->nscan == E->capscan) {
            E->capscan = E->capscan ? E->capscan * 2 : 64;
            E->scan_out = (int32_t *)realloc(E->scan_out, (size_t)E->capscan * sizeof(int32_t));
        }
        E->scan_out[E->nscan++] = id;
        return;
    }
    if (!sig_is_pending(E, id)) { E->error = 1; return; } /* ArgumentError, signal.jl:399-405 */
    cxo_value before = E->sig[id].value, v;
    if (!apply_rule(E, id, &v)) { E->error = 2; return; }
    sig_set_value(E, id, v);
    if (E->sig[id].variant == CXO_VAR_MARGINAL) E->n_marginal_updates++; else E->n_msg_updates++;
    if (E->trace_on) {
        if (E->ntrace == E->captrace) {
            E->captrace = E->captrace ? E->captrace * 2 : 256;
            E->trace = (cxo_exec *)realloc(E->trace, (size_t)E->captrace * sizeof(cxo_exec));
        }
        cxo_exec *x = &E->trace[E->ntrace++];
        x->round = E->nrounds; x->variable_id = variable_id; x->signal = id; x->before = before; x->after = v;
        E->cur_round_execs++;
    }
}

/* process_dependencies! with the closure of process_inference_request inlined
 * (signal.jl:466-490, inference_engine.jl:512-525): f(dep) = is_pending(dep) ? (process!; true) : false */
static int process_deps(cxo_engine *E, int64_t variable_id, int32_t id, int retry) {
    int any = 0;
    int32_t n = E->sig[id].ndeps;
    for (int32_t i = 0; i < n; i++) {
        int32_t dep = E->sig[id].deps[i];
        int processed = 0;
        if (sig_is_pending(E, dep)) { process(E, variable_id, dep); processed = 1; }
        if (!processed) {
            if (nib_get(&E->sig[id], i, NIB_INTERMEDIATE)) {
                int sub = process_deps(E, variable_id, dep, retry);
                if (sub && retry) {
                    if (sig_is_pending(E, dep)) { process(E, variable_id, dep); processed = 1; }
                }
                any = any || sub;
            }
        }
        any = any || processed;
    }
    return any;
}

/* ------------------------------------------------------------------ graph -- */

cxo_engine *cxo_engine_create(int32_t processor, int32_t trace) {
    cxo_engine *E = (cxo_engine *)calloc(1, sizeof(cxo_engine));
    E->processor = processor;
    E->trace_on = trace;
    E->capnodes = 64;
    E->nodes = (cxo_node *)calloc((size_t)E->capnodes, sizeof(cxo_node));
    E->nnodes = 0; /* ids are 1..nnodes; BipartiteFactorGraphs shares one id space for variables and factors
                      (test/inference_engine_tests.jl:436-443 relies on it) */
    return E;
}

static int64_t node_new(cxo_engine *E, int32_t kind) {
    if (E->nnodes + 2 > E->capnodes) {
        int64_t nc = E->capnodes * 2;
        E->nodes = (cxo_node *)realloc(E->nodes, (size_t)nc * sizeof(cxo_node));
        memset(E->nodes + E->capnodes, 0, (size_t)(nc - E->capnodes) * sizeof(cxo_node));
        E->capnodes = nc;
    }
    int64_t id = ++E->nnodes;
    E->nodes[id].kind = kind;
    E->nodes[id].marginal = -1;
    return id;
}

int64_t cxo_add_variable(cxo_engine *E) {
    int64_t id = node_new(E, 1);
    E->nodes[id].marginal = sig_new(E); /* Variable.marginal = create_inference_signal(), model_engine.jl:33 */
    return id;
}

int64_t cxo_add_factor(cxo_engine *E, int32_t fkind, double p0, double p1) {
    int64_t id = node_new(E, 2);
    E->nodes[id].fkind = fkind; E->nodes[id].p0 = p0; E->nodes[id].p1 = p1;
    return id;
}

static void nb_push(cxo_node *n, int64_t other, int32_t e) {
    if (n->nnb == n->capnb) {
        n->capnb = n->capnb ? n->capnb * 2 : 4;
        n->nb = (int64_t *)realloc(n->nb, (size_t)n->capnb * sizeof(int64_t));
        n->nbedge = (int32_t *)realloc(n->nbedge, (size_t)n->capnb * sizeof(int32_t));
    }
    /* insertion keeps the list ascending by id (the neighbour order this restatement fixes) */
    int32_t k = n->nnb++;
    while (k > 0 && n->nb[k - 1] > other) { n->nb[k] = n->nb[k - 1]; n->nbedge[k] = n->nbedge[k - 1]; k--; }
    n->nb[k] = other; n->nbedge[k] = e;
}

int32_t cxo_add_edge(cxo_engine *E, int64_t var, int64_t fac) {
    if (var < 1 || var > E->nnodes || fac < 1 || fac > E->nnodes) return -1;
    if (E->nodes[var].kind != 1 || E->nodes[fac].kind != 2) return -1;
    if (E->nedges == E->capedges) {
        E->capedges = E->capedges ? E->capedges * 2 : 64;
        E->edges = (cxo_edge *)realloc(E->edges, (size_t)E->capedges * sizeof(cxo_edge));
    }
    int32_t e = (int32_t)E->nedges++;
    E->edges[e].var = var; E->edges[e].fac = fac;
    E->edges[e].m2v = sig_new(E); E->edges[e].m2f = sig_new(E); /* Connection ctor, model_engine.jl:181-186 */
    nb_push(&E->nodes[var], fac, e);
    nb_push(&E->nodes[fac], var, e);
    return e;
}

static int32_t find_edge(cxo_engine *E, int64_t var, int64_t fac) {
    cxo_node *n = &E->nodes[var];
    int32_t lo = 0, hi = n->nnb - 1;
    while (lo <= hi) {
        int32_t mid = (lo + hi) >> 1;
        if (n->nb[mid] == fac) return n->nbedge[mid];
        if (n->nb[mid] < fac) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

int32_t cxo_message_to_variable(cxo_engine *E, int64_t var, int64_t fac) { int32_t e = find_edge(E, var, fac); return e < 0 ? -1 : E->edges[e].m2v; }
int32_t cxo_message_to_factor(cxo_engine *E, int64_t var, int64_t fac) { int32_t e = find_edge(E, var, fac); return e < 0 ? -1 : E->edges[e].m2f; }
int32_t cxo_marginal(cxo_engine *E, int64_t var) { return (var >= 1 && var <= E->nnodes && E->nodes[var].kind == 1) ? E->nodes[var].marginal : -1; }

/* dependencies.jl:128-173 (range is 0-based half-open [lo,hi) over the neighbour list here) */
static int32_t form_segment_tree(cxo_engine *E, int64_t var, int32_t lo, int32_t hi) {
    cxo_node *v = &E->nodes[var];
    int32_t len = hi - lo;
    if (len == 1) return E->edges[v->nbedge[lo]].m2v;
    int32_t mid = lo + len / 2; /* middle_point = div(length(range), 2) → left = first `len/2` */
    int32_t left = form_segment_tree(E, var, lo, mid);
    int32_t right = form_segment_tree(E, var, mid, hi);
    v = &E->nodes[var];
    for (int32_t k = lo; k < mid; k++) {
        int32_t m2f = E->edges[v->nbedge[k]].m2f;
        if (E->sig[m2f].nlist > 0) sig_add_dependency(E, m2f, right, 0, 1, 1, 1);
    }
    for (int32_t k = mid; k < hi; k++) {
        int32_t m2f = E->edges[v->nbedge[k]].m2f;
        if (E->sig[m2f].nlist > 0) sig_add_dependency(E, m2f, left, 0, 1, 1, 1);
    }
    int32_t inter = sig_new(E);
    E->sig[inter].variant = CXO_VAR_PRODUCT;
    E->sig[inter].variable_id = var;
    E->sig[inter].range_lo = lo + 1; E->sig[inter].range_hi = hi;
    sig_add_dependency(E, inter, left, 0, 1, 1, 1);
    sig_add_dependency(E, inter, right, 0, 1, 1, 1);
    return inter;
}

/* dependencies.jl:33-126 */
static void resolve_variable(cxo_engine *E, int64_t var) {
    cxo_node *v = &E->nodes[var];
    int32_t nf = v->nnb;
    int32_t marg = v->marginal;
    if (nf == 0) { /* :40-43 */
        E->warn_ctx = (int64_t *)realloc(E->warn_ctx, (size_t)(E->nwarnings + 1) * sizeof(int64_t));
        E->warn_ctx[E->nwarnings++] = var;
        return;
    }
    if (nf < 2) { /* :48-55 */
        sig_add_dependency(E, marg, E->edges[v->nbedge[0]].m2v, 0, 1, 1, 1);
        return;
    }
    if (nf <= 5) { /* :60-88 */
        for (int32_t k = 0; k < nf; k++) {
            v = &E->nodes[var];
            int32_t from = E->edges[v->nbedge[k]].m2v;
            sig_add_dependency(E, marg, from, 0, 1, 1, 1);
            int32_t m2f = E->edges[v->nbedge[k]].m2f;
            if (E->sig[m2f].nlist > 0) {
                for (int32_t j = 0; j < nf; j++) {
                    if (j == k) continue;
                    v = &E->nodes[var];
                    sig_add_dependency(E, m2f, E->edges[v->nbedge[j]].m2v, 0, 1, 1, 1);
                }
            }
        }
        return;
    }
    int32_t mid = nf / 2; /* :90-125 */
    int32_t left = form_segment_tree(E, var, 0, mid);
    int32_t right = form_segment_tree(E, var, mid, nf);
    v = &E->nodes[var];
    for (int32_t k = 0; k < mid; k++) {
        int32_t m2f = E->edges[v->nbedge[k]].m2f;
        if (E->sig[m2f].nlist > 0) sig_add_dependency(E, m2f, right, 0, 1, 1, 1);
    }
    for (int32_t k = mid; k < nf; k++) {
        int32_t m2f = E->edges[v->nbedge[k]].m2f;
        if (E->sig[m2f].nlist > 0) sig_add_dependency(E, m2f, left, 0, 1, 1, 1);
    }
    sig_add_dependency(E, marg, left, 0, 1, 1, 1);
    sig_add_dependency(E, marg, right, 0, 1, 1, 1);
}

/* InferenceEngine constructor body, inference_engine.jl:60-89 */
void cxo_engine_finalize(cxo_engine *E, int32_t resolve_dependencies) {
    E->nvars = E->nfacs = 0;
    for (int64_t id = 1; id <= E->nnodes; id++) { if (E->nodes[id].kind == 1) E->nvars++; else if (E->nodes[id].kind == 2) E->nfacs++; }
    E->var_ids = (int64_t *)malloc((size_t)(E->nvars + 1) * sizeof(int64_t));
    E->fac_ids = (int64_t *)malloc((size_t)(E->nfacs + 1) * sizeof(int64_t));
    int64_t a = 0, b = 0;
    for (int64_t id = 1; id <= E->nnodes; id++) { if (E->nodes[id].kind == 1) E->var_ids[a++] = id; else if (E->nodes[id].kind == 2) E->fac_ids[b++] = id; }
    /* set_signals_variants!, inference_engine.jl:228-247 */
    for (int64_t i = 0; i < E->nvars; i++) {
        cxo_signal *m = &E->sig[E->nodes[E->var_ids[i]].marginal];
        m->variant = CXO_VAR_MARGINAL; m->variable_id = E->var_ids[i];
    }
    for (int64_t e = 0; e < E->nedges; e++) {
        cxo_signal *s = &E->sig[E->edges[e].m2f];
        s->variant = CXO_VAR_MSG_TO_FACTOR; s->variable_id = E->edges[e].var; s->factor_id = E->edges[e].fac;
        s = &E->sig[E->edges[e].m2v];
        s->variant = CXO_VAR_MSG_TO_VARIABLE; s->variable_id = E->edges[e].var; s->factor_id = E->edges[e].fac;
    }
    if (!resolve_dependencies) return;
    /* resolve_dependencies!, dependencies.jl:5-15: factors first, then variables */
    for (int64_t i = 0; i < E->nfacs; i++) { /* dependencies.jl:17-31 */
        int64_t f = E->fac_ids[i];
        int32_t n = E->nodes[f].nnb;
        for (int32_t i1 = 0; i1 < n; i1++)
            for (int32_t i2 = 0; i2 < n; i2++) {
                if (i1 == i2) continue;
                cxo_node *fn = &E->nodes[f];
                sig_add_dependency(E, E->edges[fn->nbedge[i1]].m2v, E->edges[fn->nbedge[i2]].m2f, 0, 1, 1, 0);
            }
    }
    for (int64_t i = 0; i < E->nvars; i++) resolve_variable(E, E->var_ids[i]);
}

/* ------------------------------------------------------------- scheduler -- */

/* request_inference_for, inference_engine.jl:298-323 */
static void request_inference(cxo_engine *E, const int64_t *ids, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        cxo_signal *m = &E->sig[E->nodes[ids[i]].marginal];
        for (int32_t k = 0; k < m->ndeps; k++) {
            cxo_signal *d = &E->sig[m->deps[k]];
            d->potentially_pending = 1; d->pending = 0;
        }
        for (int32_t k = 0; k < E->nodes[ids[i]].nlinked; k++) { /* :313-317 */
            cxo_signal *d = &E->sig[E->nodes[ids[i]].linked[k]];
            d->potentially_pending = 1; d->pending = 0;
        }
    }
}

/* scan_inference_request, inference_engine.jl:540-546 */
int64_t cxo_scan(cxo_engine *E, const int64_t *ids, int64_t n, int32_t *out, int64_t cap) {
    request_inference(E, ids, n);
    E->scanning = 1; E->nscan = 0;
    for (int64_t i = 0; i < n; i++) process_deps(E, ids[i], E->nodes[ids[i]].marginal, 1);
    E->scanning = 0;
    for (int64_t i = 0; i < E->nscan && i < cap; i++) out[i] = E->scan_out[i];
    return E->nscan;
}

static void round_end(cxo_engine *E) { if (E->trace_on && E->cur_round_execs > 0) E->nrounds++; E->cur_round_execs = 0; }

/* update_marginals!, inference_engine.jl:559-632 */
int32_t cxo_update_marginals(cxo_engine *E, const int64_t *ids, int64_t n) {
    E->error = 0;
    E->ntrace = 0; E->nrounds = 0; E->cur_round_execs = 0;
    request_inference(E, ids, n);
    uint8_t *ready = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    int should_continue = 1, is_reverse = 0;
    while (should_continue) {
        int cont = 0;
        for (int64_t k = 0; k < n; k++) {
            int64_t i = is_reverse ? (n - 1 - k) : k;
            if (!ready[i]) {
                int32_t marg = E->nodes[ids[i]].marginal;
                int did = process_deps(E, ids[i], marg, 1);
                if (sig_is_pending(E, marg)) ready[i] = 1;
                cont = cont || did;
                if (E->error) { free(ready); return E->error; }
            }
        }
        round_end(E);
        is_reverse = !is_reverse;
        should_continue = cont;
    }
    for (int64_t i = 0; i < n; i++) { /* final round :610-628 */
        int32_t marg = E->nodes[ids[i]].marginal;
        if (sig_is_pending(E, marg)) process(E, ids[i], marg);
        if (E->error) { free(ready); return E->error; }
        for (int32_t k = 0; k < E->nodes[ids[i]].nlinked; k++) { /* linked signals, :618-626 */
            int32_t ls = E->nodes[ids[i]].linked[k];
            if (!sig_is_pending(E, ls)) continue;
            process(E, ids[i], ls);
            if (E->error) { free(ready); return E->error; }
        }
    }
    round_end(E);
    free(ready);
    return 0;
}

/* --------------------------------------------------------- raw accessors -- */

int32_t cxo_signal_new(cxo_engine *E) { return sig_new(E); }
void cxo_add_dependency(cxo_engine *E, int32_t s, int32_t d, int32_t weak, int32_t listen, int32_t check_computed, int32_t intermediate) { sig_add_dependency(E, s, d, weak, listen, check_computed, intermediate); }
void cxo_set_value(cxo_engine *E, int32_t s, int32_t tag, double a, double b) { cxo_value v = { tag, a, b, { 0, 0, 0, 0 } }; sig_set_value(E, s, v); }
void cxo_set_value_ex(cxo_engine *E, int32_t s, int32_t tag, const double *six) {
    cxo_value v = { tag, six[0], six[1], { six[2], six[3], six[4], six[5] } };
    sig_set_value(E, s, v);
}
int32_t cxo_get_value_ex(cxo_engine *E, int32_t s, double *six) {
    cxo_value v = E->sig[s].value;
    six[0] = v.a; six[1] = v.b;
    for (int k = 0; k < 4; k++) six[2 + k] = v.x[k];
    return v.tag;
}
/* resolve_variable_dependencies!(DefaultDependencyResolver(), ...) for one variable: what a custom resolver delegates to
 * (test/inference_engine_tests.jl:812-814) */
void cxo_resolve_variable_default(cxo_engine *E, int64_t var) { resolve_variable(E, var); }
void cxo_set_rule_callback(cxo_engine *E, cxo_rule_cb cb, void *ctx) { E->rule_cb = cb; E->rule_ctx = ctx; }
/* link_signal_to_variable!, model_engine.jl:75-83 */
void cxo_link_signal_to_variable(cxo_engine *E, int64_t var, int32_t s) {
    cxo_node *n = &E->nodes[var];
    if (n->nlinked == n->caplinked) {
        n->caplinked = n->caplinked ? n->caplinked * 2 : 4;
        n->linked = (int32_t *)realloc(n->linked, (size_t)n->caplinked * sizeof(int32_t));
    }
    n->linked[n->nlinked++] = s;
}
/* set_variant! for signals the resolver creates itself (JointMarginal(factor_id, variable_ids), inference_signal.jl:93-96):
 * variable_id carries the first variable of the cluster, range_lo/hi the first and last */
void cxo_set_variant(cxo_engine *E, int32_t s, int32_t variant, int64_t variable_id, int64_t factor_id, int32_t lo, int32_t hi) {
    cxo_signal *g = &E->sig[s];
    g->variant = variant; g->variable_id = variable_id; g->factor_id = factor_id; g->range_lo = lo; g->range_hi = hi;
}
int32_t cxo_is_pending(cxo_engine *E, int32_t s) { return sig_is_pending(E, s); }
int32_t cxo_is_computed(cxo_engine *E, int32_t s) { return sig_is_computed(E, s); }
int32_t cxo_get_value(cxo_engine *E, int32_t s, double *a, double *b) { *a = E->sig[s].value.a; *b = E->sig[s].value.b; return E->sig[s].value.tag; }
int32_t cxo_num_dependencies(cxo_engine *E, int32_t s) { return E->sig[s].ndeps; }
int32_t cxo_dependency(cxo_engine *E, int32_t s, int32_t i) { return E->sig[s].deps[i]; }
int32_t cxo_num_listeners(cxo_engine *E, int32_t s) { return E->sig[s].nlist; }
int32_t cxo_listener(cxo_engine *E, int32_t s, int32_t i) { return E->sig[s].listeners[i]; }
int32_t cxo_num_chunks(cxo_engine *E, int32_t s) { return E->sig[s].nchunks; }
uint64_t cxo_chunk(cxo_engine *E, int32_t s, int32_t c) { return *chunk_ptr(&E->sig[s], c); }
int32_t cxo_variant(cxo_engine *E, int32_t s, int64_t *var, int64_t *fac, int32_t *lo, int32_t *hi) {
    cxo_signal *x = &E->sig[s]; *var = x->variable_id; *fac = x->factor_id; *lo = x->range_lo; *hi = x->range_hi; return x->variant;
}
int64_t cxo_num_signals(cxo_engine *E) { return E->nsig; }
int32_t cxo_num_warnings(cxo_engine *E) { return E->nwarnings; }
int64_t cxo_warning_context(cxo_engine *E, int32_t i) { return E->warn_ctx[i]; }
int64_t cxo_num_variables(cxo_engine *E) { return E->nvars; }
int64_t cxo_num_factors(cxo_engine *E) { return E->nfacs; }
void cxo_variable_ids(cxo_engine *E, int64_t *out) { memcpy(out, E->var_ids, (size_t)E->nvars * sizeof(int64_t)); }
void cxo_factor_ids(cxo_engine *E, int64_t *out) { memcpy(out, E->fac_ids, (size_t)E->nfacs * sizeof(int64_t)); }
int32_t cxo_num_neighbors(cxo_engine *E, int64_t id) { return E->nodes[id].nnb; }
int64_t cxo_neighbor(cxo_engine *E, int64_t id, int32_t k) { return E->nodes[id].nb[k]; }
int64_t cxo_trace_len(cxo_engine *E) { return E->ntrace; }
int64_t cxo_trace_rounds(cxo_engine *E) { return E->nrounds; }
void cxo_trace_get(cxo_engine *E, int64_t i, int64_t *round, int64_t *variable_id, int32_t *signal,
                   int32_t *tag_before, double *a_before, int32_t *tag_after, double *a_after, double *b_after) {
    cxo_exec *x = &E->trace[i];
    *round = x->round; *variable_id = x->variable_id; *signal = x->signal;
    *tag_before = x->before.tag; *a_before = x->before.a; *tag_after = x->after.tag; *a_after = x->after.a; *b_after = x->after.b;
}
int64_t cxo_counter(cxo_engine *E, int32_t which) { return which == 0 ? E->n_msg_updates : E->n_marginal_updates; }
int32_t cxo_last_error(cxo_engine *E) { return E->error; }

/* process_dependencies! with a user callback, for the signal_tests.jl:933-1082 known answers */
typedef int32_t (*cxo_cb)(int32_t sig, void *ctx);
static int pd_cb(cxo_engine *E, int32_t id, int retry, cxo_cb f, void *ctx) {
    int any = 0;
    for (int32_t i = 0; i < E->sig[id].ndeps; i++) {
        int32_t dep = E->sig[id].deps[i];
        int processed = f(dep, ctx);
        if (!processed && nib_get(&E->sig[id], i, NIB_INTERMEDIATE)) {
            int sub = pd_cb(E, dep, retry, f, ctx);
            if (sub && retry) processed = f(dep, ctx);
            any = any || sub;
        }
        any = any || processed;
    }
    return any;
}
int32_t cxo_process_dependencies(cxo_engine *E, int32_t s, int32_t retry, cxo_cb f, void *ctx) { return pd_cb(E, s, retry, f, ctx); }

/* compute!(strategy, signal; force, skip_if_no_listeners) on a free signal, signal.jl:392-410.  The strategy sees the signal and its
 * dependencies and returns a value.  0: computed and stored (set_value!), 3: skipped — no listeners, 1: the ArgumentError of a
 * non-pending signal without force, 2: the strategy failed. */
typedef int32_t (*cxo_strategy_cb)(void *ctx, int32_t sig, int32_t ndeps, const int32_t *deps, int32_t *tag_out, double *a_out, double *b_out);
int32_t cxo_compute(cxo_engine *E, int32_t s, int32_t force, int32_t skip_if_no_listeners, cxo_strategy_cb strategy, void *ctx) {
    if (skip_if_no_listeners && E->sig[s].nlist == 0) return 3;
    if (!force && !sig_is_pending(E, s)) return 1;
    cxo_value v = { CXO_UNDEF, 0.0, 0.0, { 0, 0, 0, 0 } };
    if (!strategy(ctx, s, E->sig[s].ndeps, E->sig[s].deps, &v.tag, &v.a, &v.b)) return 2;
    sig_set_value(E, s, v);
    return 0;
}

/* bulk helpers so Python never loops over 1e6 signals */
void cxo_bulk_set_message_to_factor(cxo_engine *E, const int64_t *vars, const int64_t *facs, int64_t n, int32_t tag, const double *a, const double *b) {
    for (int64_t i = 0; i < n; i++) { cxo_value v = { tag, a[i], b ? b[i] : 0.0, { 0, 0, 0, 0 } }; sig_set_value(E, cxo_message_to_factor(E, vars[i], facs[i]), v); }
}
void cxo_bulk_set_message_to_variable(cxo_engine *E, const int64_t *vars, const int64_t *facs, int64_t n, int32_t tag, const double *a, const double *b) {
    for (int64_t i = 0; i < n; i++) { cxo_value v = { tag, a[i], b ? b[i] : 0.0, { 0, 0, 0, 0 } }; sig_set_value(E, cxo_message_to_variable(E, vars[i], facs[i]), v); }
}
void cxo_bulk_get_marginals(cxo_engine *E, const int64_t *vars, int64_t n, int32_t *tags, double *a, double *b) {
    for (int64_t i = 0; i < n; i++) { cxo_value v = E->sig[E->nodes[vars[i]].marginal].value; tags[i] = v.tag; a[i] = v.a; b[i] = v.b; }
}
void cxo_bulk_get_messages(cxo_engine *E, const int64_t *vars, const int64_t *facs, int64_t n, int32_t to_variable, int32_t *tags, double *a, double *b) {
    for (int64_t i = 0; i < n; i++) {
        int32_t s = to_variable ? cxo_message_to_variable(E, vars[i], facs[i]) : cxo_message_to_factor(E, vars[i], facs[i]);
        cxo_value v = E->sig[s].value; tags[i] = v.tag; a[i] = v.a; b[i] = v.b;
    }
}
/* bulk graph construction: n_nodes ids, kind[id-1] ∈ {1 var, 2 factor} */
int32_t cxo_bulk_build(cxo_engine *E, int64_t n_nodes, const int32_t *kind, const int32_t *fkind, const double *p0,
                       int64_t n_edges, const int64_t *evar, const int64_t *efac) {
    for (int64_t i = 0; i < n_nodes; i++) {
        if (kind[i] == 1) cxo_add_variable(E); else cxo_add_factor(E, fkind[i], p0[i], 0.0);
    }
    for (int64_t e = 0; e < n_edges; e++) if (cxo_add_edge(E, evar[e], efac[e]) < 0) return -1;
    return 0;
}

void cxo_engine_destroy(cxo_engine *E) {
    if (!E) return;
    for (int64_t i = 0; i < E->nsig; i++) { free(E->sig[i].deps); free(E->sig[i].xchunks); free(E->sig[i].listeners); free(E->sig[i].listenmask); }
    for (int64_t i = 1; i <= E->nnodes; i++) { free(E->nodes[i].nb); free(E->nodes[i].nbedge); free(E->nodes[i].linked); }
    free(E->sig); free(E->nodes); free(E->edges); free(E->var_ids); free(E->fac_ids); free(E->warn_ctx); free(E->trace); free(E->scan_out);
    free(E);
}
