/*
 * cortex_hip.h — C ABI of libcortex_hip.so: the MI355X (gfx950) sum-product sweep that sits
 * behind Cortex.jl's InferenceEngine / AbstractInferenceRequestProcessor plugin API.
 *
 * The reference (Cortex.jl v0.3.0, pure Julia) has no FFI; these entry points are what a Julia
 * `ccall` shim for this path binds (INTEGRATION.md shows the stub).  Each declaration names the
 * reference interface it replaces (paths relative to the reference checkout):
 *
 *   graph ingestion      the 7 model-engine accessors, src/model_engine.jl:329-391, as forwarded by
 *                        ext/BipartiteFactorGraphsExt/BipartiteFactorGraphsExt.jl:16-48
 *   cx_set_messages      set_value!(message, data)                         src/signal.jl:232-253
 *   cx_update_batch      process!(processor, engine, variable_id, signal)  src/inference_engine.jl:479-509
 *                        → compute_message_to_variable! / compute_message_to_factor! /
 *                          compute_individual_marginal! / compute_product_of_messages! /
 *                          compute_joint_marginal!                         src/inference_engine.jl:351-477
 *   cx_sweep             update_marginals!(engine, variable_ids)           src/inference_engine.jl:559-632
 *                        (device "flooding" schedule over the whole graph)
 *   cx_get_marginals     get_value(get_variable_marginal(variable))        src/model_engine.jl:60-62
 *   cx_get_messages      get_value(message_to_variable / message_to_factor) src/model_engine.jl:207-222
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns an int32 status
 *     (CX_OK == 0, negative = error) and never throws or longjmps across the boundary;
 *     cx_last_error() returns the text a Julia shim passes to error().
 *   - ids are the reference's ids: 1-based Int64, one id space shared by variables and factors
 *     (BipartiteFactorGraphs add_variable!/add_factor!).  Neighbour order is ascending id.
 *   - Gaussian payloads cross the boundary in MOMENT form like the reference's test structs
 *     (test/runtests.jl:31-34 NormalMeanVariance): dim==1: {mean, variance};
 *     dim==d: mean[d] then covariance[d*d] row-major.  NaN variance/covariance == UndefValue().
 *   - the caller owns every pointer it passes; the library copies before returning and owns all
 *     device memory behind the opaque handle.  One caller thread per handle.
 *   - calls are asynchronous on the handle's HIP stream unless they return data to the host;
 *     cx_sync() waits.
 */
#ifndef CORTEX_HIP_H
#define CORTEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CX_ABI_VERSION 4   /* 4: cx_config.sweeps_per_launch -> reserved, cx_tile_stats and CX_KERNEL_TILED removed.  3: CX_SCHED_REFERENCE, cx_sweep_for, cx_ref_plan_stats, cx_ref_trace, cx_set_damping.  2: cx_config.reserved became sweeps_per_launch (validated), five new item / factor kinds, state blobs "CXSTATE2" */

/* status codes */
#define CX_OK 0
#define CX_ERR_INVALID_ARGUMENT (-1)  /* bad pointer / size / enum */
#define CX_ERR_NOT_FOUND (-2)         /* unknown variable / factor / edge id */
#define CX_ERR_UNSUPPORTED (-3)       /* factor arity / kind / dim the device path does not implement */
#define CX_ERR_STATE (-4)             /* call order (e.g. sweep before graph) */
#define CX_ERR_DEVICE (-5)            /* HIP runtime error; text in cx_last_error */
#define CX_ERR_NO_DEVICE (-6)         /* no gfx950 device visible */
#define CX_ERR_OUT_OF_MEMORY (-7)

/* message direction: which of a Connection's two signals (model_engine.jl:181-186) */
#define CX_TO_FACTOR 1    /* MessageToFactor(variable_id, factor_id),   inference_signal.jl:29-32 */
#define CX_TO_VARIABLE 2  /* MessageToVariable(variable_id, factor_id), inference_signal.jl:45-48 */

/* work-item kinds of cx_update_batch == the variants process! dispatches on */
#define CX_ITEM_MESSAGE_TO_FACTOR 1
#define CX_ITEM_MESSAGE_TO_VARIABLE 2
#define CX_ITEM_INDIVIDUAL_MARGINAL 4
#define CX_ITEM_PRODUCT_OF_MESSAGES 8  /* ProductOfMessages(variable_id, range, factors), inference_signal.jl:62-66: the product of
                                          the factor→variable messages number lo..hi (1-based, inclusive, ascending factor id:
                                          the reference's `range` over factors_connected_to_variable) of variable_id — the
                                          segment-tree intermediates of dependencies.jl:128-173.  ORDERING ASSUMPTION: the range is
                                          resolved over the variable's factors in ascending id order; a plug-in whose model engine
                                          lists factors_connected_to_variable in another order must refuse or translate (the Python
                                          and Julia plug-ins check it).  The range travels in
                                          cx_item.factor_id as CX_ITEM_RANGE(lo, hi); the value is kept in a device store
                                          (cx_get_products) */
#define CX_ITEM_JOINT_MARGINAL 16      /* JointMarginal(factor_id, variable_ids), inference_signal.jl:93-96, for a pairwise
                                          Gaussian factor: the 2-d Gaussian ∝ factor x the two variable→factor messages
                                          (compute_joint_marginal!, inference_engine.jl:469-477); variable_id is ignored, the
                                          value is kept in a device store (cx_get_joint_marginals) */
#define CX_ITEM_RANGE(lo, hi) ((int64_t)(((uint64_t)(uint32_t)(lo) << 32) | (uint64_t)(uint32_t)(hi)))

/* payload forms */
#define CX_FORM_MOMENT 0  /* Gaussian (mean, variance|covariance) */
#define CX_FORM_POINT 1   /* observed datum: the `Real` input of test/inference_engine_tests.jl:424 ; dim values */
#define CX_FORM_NATURAL 2 /* (xi = precision*mean, w = precision): the library's storage form */
#define CX_FORM_MEAN_PRECISION 3 /* NormalMeanPrecision(mean, precision), test/runtests.jl:48-56 (variational families) */
#define CX_FORM_GAMMA 4          /* Gamma(shape, scale), test/runtests.jl:60-67 (variational families) */

/* factor kinds (what the user rule reads from Factor.functional_form, model_engine.jl:119-122) */
#define CX_FACTOR_OPAQUE 0          /* messages only ever set by the caller (priors, unary observations) */
#define CX_FACTOR_GAUSS_ADDITIVE 1  /* 2 edges: x_b = x_a + N(0, q)            params = {q}                */
#define CX_FACTOR_GAUSS_LINEAR 2    /* 2 edges: x_out = a*x_in + b + N(0, q)   params = {q, a, b}          */
                                    /* dim>1:   x_out = A x_in + N(0, Q)       params via cx_set_factor_matrices */
#define CX_FACTOR_NORMAL_PRECISION 3 /* 3 edges: x_out ~ N(x_in, 1 / precision); roles OUT, IN, PRECISION — the :likelihood and
                                       :transition factors of test/inference_engine_tests.jl:691-715.  Variational rules only: the
                                       CX_FAMILY_VMP_* families, or CX_FAMILY_GAUSSIAN under CX_SCHED_REFERENCE with a user wiring
                                       (cx_graph_wire, ABI 3) */
#define CX_FACTOR_BERNOULLI 4        /* CX_FAMILY_NATURAL2, 2 edges: one variable carries an observed Bool r (a CX_FORM_POINT datum),
                                       the message to the other is Beta(1 + r, 2 - r) = natural (r, 1 - r): the :bernoulli
                                       factor of test/inference_engine_tests.jl:250-262.  Without a datum the reference's rule
                                       is error(...): the message stays undefined */
#define CX_FACTOR_GAUSS_LINEAR_N 5   /* 3 to 7 edges.  dim 2, 3, 4 (ABI 3): x_out = A_1 x_1 + ... + A_k x_k + N(0, Q), params = {parameter set}: its Q is the noise, its A
                                        every input's matrix unless cx_set_factor_edge_sets names another set for an edge; fused and tree schedules,
                                        batch items.  dim == 1: x_out = a_1 x_1 + ... + a_k x_k + b + N(0, q), k = 2..6 inputs; params = {q, b};
                                        exactly one CX_ROLE_OUT edge, the others CX_ROLE_IN; a_i = 1 unless cx_set_factor_coefficients says
                                        otherwise.  Every factor→variable message of the factor reads ALL its other variable→factor
                                        messages (src/dependencies.jl:17-31).  Flooding, fused, tree and reference-order schedules; under partitions
                                        with state halos (cx_halo_configure_state), not with per-sweep message halos. */
#define CX_NPARAM 4                 /* doubles per factor in factor_params */

/* edge roles for directed factors (Connection.label :out/:in, model_engine.jl:182) */
#define CX_ROLE_OUT 0
#define CX_ROLE_IN 1
#define CX_ROLE_PRECISION 2   /* the Gamma-distributed precision of a CX_FACTOR_NORMAL_PRECISION factor */

/* message families for dim == 1.  The sweep's products are sums of natural parameters for ANY exponential family; only
 * the factor rules, the moment conversions at the ABI and the marginal read-out are Gaussian-specific. */
#define CX_FAMILY_GAUSSIAN 0  /* (xi, w); marginals and MOMENT payloads are (mean, variance) */
#define CX_FAMILY_NATURAL2 1  /* any 2-parameter family in natural coordinates, e.g. Beta(a, b) as (a-1, b-1): the
                                 Beta-Bernoulli model of test/inference_engine_tests.jl:241-377.  Payloads are NATURAL (data:
                                 POINT), factors are CX_FACTOR_OPAQUE (their messages are set by the caller) or
                                 CX_FACTOR_BERNOULLI, marginals come back as the natural-parameter sums. */
/* Variational message passing: messages depend WEAKLY on marginals (add_dependency!(...; weak = true), signal.jl:36-45).
 * The state is the set of marginals (cx_set_marginals / cx_update_marginals / cx_get_marginals); cx_sweep, cx_update_batch
 * and the message accessors do not apply.  Factors are CX_FACTOR_NORMAL_PRECISION, dim == 1. */
#define CX_FAMILY_VMP_MEAN_FIELD 2  /* fully factorised posterior: the MeanFieldResolver + SSMMeanFieldInferenceRequestProcessor
                                       of test/inference_engine_tests.jl:599-689 */
#define CX_FAMILY_VMP_STRUCTURED 3  /* Normal variables jointly (belief propagation with E[precision], run with cfg.schedule —
                                       CX_SCHED_CHAIN_SCAN gives the exact forward/backward pass per update, CX_SCHED_TREE the exact
                                       two passes when the states form any forest), precisions from the joint marginals:
                                       StructuredResolver + SSMStructuredInferenceRequestProcessor, :810-1030 */

/* schedules of cx_sweep */
#define CX_SCHED_FLOODING 0   /* all variable→factor, then all factor→variable, then marginals           */
#define CX_SCHED_FUSED 1      /* same fixed-point map, one fused kernel on double-buffered messages        */
#define CX_SCHED_CHAIN_SCAN 2 /* graphs whose non-observed variables form disjoint chains (state-space models):
                                 one cx_sweep = the exact forward/backward result of the reference's sequential
                                 schedule (inference_engine.jl:575-608), by two parallel prefix scans.  When every
                                 non-observed variable is on a chain and materialize_messages_to_factor is 0, the
                                 scan writes the marginals itself and variable→factor messages are recomputed from
                                 the stored messages when cx_get_messages / cx_update_batch ask for them.
                                 dim 1, 2, 3, 4.  For dim 2..4 (scan over composed linear-Gaussian maps) every
                                 non-observed variable must sit on a chain; a sweep writes the marginals, the chain
                                 messages reach their slots when cx_get_messages / cx_update_batch / cx_residual /
                                 cx_state_export ask for them                                                     */
#define CX_SCHED_TREE 3       /* graphs whose non-observed part is a forest (any degrees, factors of two or more variables):
                                 one cx_sweep = the reference's one update_marginals! there — every message once, from
                                 final inputs, leaves to root and back (inference_engine.jl:575-608), level by level:
                                 2 x depth + 1 launches over item lists built once (depth = half the longest path of a
                                 component, counted in variables and factors).  A cycle among the non-observed variables
                                 is refused (CX_ERR_UNSUPPORTED).  Lazy like the reference: nothing into observed
                                 variables.  dim 1 (Gaussian and natural-pair families), 2, 3, 4 and 64 (5 .. 63 with it).  */
#define CX_SCHED_REFERENCE 4  /* ANY graph, loops included: one cx_sweep / cx_sweep_for = the reference's one update_marginals! —
                                 the same signals computed in the same ORDER, each from exactly the values the reference's rule call
                                 read (a sequential pass that always reads the newest values, inference_engine.jl:575-608,
                                 signal.jl:466-490: "Gauss-Seidel" in requested-variable order x neighbour order).  The library keeps a
                                 shadow of every signal's readiness nibbles, wired as DefaultDependencyResolver wires them
                                 (dependencies.jl:17-173, signal.jl:141-154,232-253,668-730) and driven by cx_set_messages /
                                 cx_seed_messages (the user's set_value!), cx_update_batch (a plug-in's process!) and the sweeps; a call
                                 runs the reference's scheduler on the shadow, records the executions, levels them (an execution's stage
                                 follows every execution whose result it reads, every reader of the value it overwrites, and its own
                                 previous one) and replays the stages as item lists in ONE launch: an XCD-resident cluster — the
                                 workgroups of one XCD behind barriers that stay in that XCD's L2 — for plans of wide stages (such a call
                                 returns when it has finished: the host checks that no barrier timed out), a HIP graph of stage launches
                                 for chains of thin ones (asynchronous).  Plans are kept per (readiness
                                 state at the start of the call, request): the steady state of "set the priors, call" replays a standing
                                 plan.  Lazy like the reference: a call computes what is pending for the requested marginals, nothing
                                 else; priors have to be re-set before a call to be fresh, exactly as there.  On a forest it is the tree
                                 schedule with requests for some of the variables.  dim 2, 3, 4 (ABI 3): linear-Gaussian factors of two to seven
                                 variables, variables of any degree, the default wiring (the stages run through the batched d-dimensional items;
                                 variable→factor messages are stored signals there, not recomputed for a reader).  dim 1 (Gaussian and natural-pair families), factors
                                 of any arity the rules have, variables of any degree (degree > 5: the segment-tree nodes of
                                 dependencies.jl:90-173 are computed, stored and read one by one, as the reference does).  Not partitioned. */


typedef struct cx_handle cx_handle;

typedef struct cx_config {
    int32_t struct_size;   /* sizeof(cx_config), for forward compatibility */
    int32_t device;        /* HIP device ordinal */
    int32_t dim;           /* message dimension d: 1 (scalar), 2, 3, 4 (registers); 5 .. 64 run on the f64 matrix cores, a wave per message, in
                              1 x 1, 2 x 2 or 4 x 4 accumulator tiles of 16: the smallest of 16, 32, 64 that holds d, under every schedule
                              (ABI 4; cx_chain_block_maps: dim 1 .. 4 and 64 only).  A d in between is embedded
                              block-diagonally (x, u) with u a unit random walk nobody observes: exact results for the d x d blocks,
                              payloads of d and d + d*d doubles as for any d, at the cost of the tile size                          */
    int32_t schedule;      /* CX_SCHED_*: dim 1 all five; dim 2 .. 64: CX_SCHED_FUSED, CX_SCHED_CHAIN_SCAN, CX_SCHED_TREE, CX_SCHED_REFERENCE */
    int32_t compute_marginals_in_sweep; /* 1: every sweep also refreshes all marginals (update_marginals!).
                                           2 (CX_SCHED_CHAIN_SCAN, dim 2..4; elsewhere the same as 1): on demand — a sweep leaves
                                           the chain's forward and backward sums in the order of its walks, and the pass that
                                           adds them up, converts to moment form and moves them to the marginals' place runs
                                           before the first cx_state_export / cx_update_batch / cx_get_marginals of an eighth of
                                           the variables or more after it; a cx_get_marginals for fewer forms just those
                                           (the same values; dim 64 always works this way)                                  */
    int32_t materialize_messages_to_factor; /* CX_SCHED_FUSED only. 0: variable→factor messages stay in registers
                                               during a sweep and are recomputed, bit-identically, from the retained
                                               input buffer when cx_get_messages / cx_update_batch asks for them;
                                               1: every sweep also stores them */
    int32_t family;        /* CX_FAMILY_*: what a scalar (dim == 1) message's two numbers mean */
    int32_t reserved;      /* 0 (1 is accepted).  ABI 2 - 3: sweeps_per_launch — the two-sweeps-per-launch kernel (temporal blocking in
                              LDS; bit-identical, measured 1.8 - 2x slower than two single sweeps, HISTORY.md) was removed in ABI 4 */
} cx_config;

typedef struct cx_item {
    int32_t kind;          /* CX_ITEM_* */
    int32_t reserved;
    int64_t variable_id;
    int64_t factor_id;     /* ignored for CX_ITEM_INDIVIDUAL_MARGINAL; CX_ITEM_RANGE(lo, hi) for CX_ITEM_PRODUCT_OF_MESSAGES */
} cx_item;

typedef struct cx_stats {
    int64_t n_variables, n_factors, n_edges;
    int64_t n_messages_per_sweep;     /* directed messages with >=1 dependency and >=1 listener: the metric's unit */
    int64_t n_slices, n_big_variables, n_slots; /* SELL-256 slices, CSR-tail variables, message slots incl. padding */
    int64_t device_bytes;
    int64_t sweeps_done;
} cx_stats;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int32_t cx_version(void);
int32_t cx_create(const cx_config *config, cx_handle **out);
int32_t cx_destroy(cx_handle *h);                 /* idempotent on NULL */
const char *cx_last_error(const cx_handle *h);    /* h may be NULL: last error of a failed cx_create */
int32_t cx_sync(cx_handle *h);
int32_t cx_set_stream(cx_handle *h, void *hip_stream); /* run on the caller's hipStream_t (NULL = default) */

/* ---- graph ingestion: BipartiteFactorGraph{Variable,Factor,Connection} flattened once --------
 * edge_var[e], edge_fac[e]: the (variable_id, factor_id) of every Connection, any order.
 * factor_ids[f], factor_kind[f], factor_params[f*CX_NPARAM ..]: one row per factor.
 * edge_role may be NULL (all CX_ROLE_OUT); it is only read for CX_FACTOR_GAUSS_LINEAR.
 * Replaces get_variable_ids/get_factor_ids/get_connected_*_ids/get_connection
 * (model_engine.jl:329-391) and fixes the dependency wiring of DefaultDependencyResolver
 * (dependencies.jl:17-126) into gather lists. */
int32_t cx_graph_create(cx_handle *h, int64_t n_edges, const int64_t *edge_var, const int64_t *edge_fac,
                        const int32_t *edge_role, int64_t n_factors, const int64_t *factor_ids,
                        const int32_t *factor_kind, const double *factor_params);
/* dim > 1: parameter set `parameter_set` of the linear-Gaussian factors x_out = A x_in + N(0, Q); A, Q are d x d row-major,
 * Q symmetric positive definite.  A CX_FACTOR_GAUSS_LINEAR factor names its set in factor_params[0]; its ROLE_IN edge carries
 * x_in.  (The reference leaves Factor.functional_form opaque, model_engine.jl:119-122; this is what the user rule reads.) */
int32_t cx_set_factor_matrices(cx_handle *h, int64_t parameter_set, const double *A, const double *Q);
/* a_i of ROLE_IN edges of CX_FACTOR_GAUSS_LINEAR_N factors (finite, non-zero; default 1).  Takes effect at the next sweep. */
int32_t cx_set_factor_coefficients(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, const double *a);
/* dim 2, 3, 4: x_out = A_1 x_1 + ... + A_k x_k + N(0, Q) for a CX_FACTOR_GAUSS_LINEAR_N factor: the factor's params[0] names the parameter set
 * whose Q is the noise and whose A is every input's matrix; this call gives the CX_ROLE_IN edge (variable, factor) the A of another set. */
int32_t cx_set_factor_edge_sets(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids, const int64_t *parameter_sets);
int32_t cx_graph_stats(const cx_handle *h, cx_stats *out);
/* position of Connection (variable_id, factor_id) in the flattened edge table (sorted by variable, factor) */
int32_t cx_edge_index(const cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids,
                      int64_t *out_edge);

/* ---- data injection / read-back --------------------------------------------------------------
 * payload: n rows of cx_payload_doubles(dim, form) doubles. */
int64_t cx_payload_doubles(int32_t dim, int32_t form);
int32_t cx_set_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids,
                        int32_t direction, int32_t form, const double *payload);
int32_t cx_get_messages(cx_handle *h, int64_t n, const int64_t *variable_ids, const int64_t *factor_ids,
                        int32_t direction, int32_t form, double *out);
/* ---- variational families: the marginals are the state -------------------------------------------------------------
 * cx_set_marginals    : the user's set_value!(get_variable_marginal(...), value) (test/inference_engine_tests.jl:717-727,
 *                       :734-736): CX_FORM_POINT observes a Normal variable (1 double), CX_FORM_MEAN_PRECISION /
 *                       CX_FORM_MOMENT initialise a latent Normal variable, CX_FORM_GAMMA a precision variable (2 doubles).
 * cx_update_marginals : one update_marginals!(engine, variable_ids) (src/inference_engine.jl:559-632) under the weak
 *                       wiring: the messages into the requested variables are computed from the marginals as they stand
 *                       before the call, then the requested marginals are stored.  n may be CX_VMP_ALL_NORMAL or
 *                       CX_VMP_ALL_PRECISION (variable_ids ignored) to name a whole class without an id list.
 *                       CX_FAMILY_VMP_STRUCTURED updates the latent Normal variables together.  A request that names them TOGETHER
 *                       with precision variables (the last call of the reference's experiment, test/inference_engine_tests.jl:1113)
 *                       is evaluated by the reference in an order that emerges from its lazy readiness flags; it is accepted where
 *                       that order is class by class — transition precisions, states, other precisions — i.e. when the states were
 *                       updated before, every precision of a transition factor (degree > 5) precedes the first state in the list and
 *                       every other requested precision was updated since the states last were; CX_ERR_UNSUPPORTED otherwise (the
 *                       reference then interleaves per variable).  Asynchronous on the handle's stream.
 * cx_get_marginals    : (mean, precision) for Normal variables ((datum, +inf) when observed), (shape, scale) for precisions.
 * cx_set_marginals also serves CX_SCHED_REFERENCE handles (dim 1) whose wiring makes messages depend on marginals (cx_graph_wire): the
 * same forms; there cx_get_marginals returns (mean, variance) for Normal variables — the Gaussian family's form — and (shape, scale) for
 * precisions. */
#define CX_VMP_ALL_NORMAL (-1)
#define CX_VMP_ALL_PRECISION (-2)
int32_t cx_set_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, int32_t form, const double *payload);
int32_t cx_update_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids);

/* give every still-undefined message of `direction` the value N(mean, variance*I): the seeding a user of
 * the reference does by hand before loopy BP (cf. test/inference_engine_tests.jl:729-736) */
int32_t cx_seed_messages(cx_handle *h, int32_t direction, double mean, double variance);
int32_t cx_get_marginals(cx_handle *h, int64_t n, const int64_t *variable_ids, double *out);

/* ---- compute ---------------------------------------------------------------------------------- */
/* one launch for a batch of mutually independent signals, in the caller's order of enqueue.  dim == 1: all five item kinds;
 * dim 2, 3, 4, 64: CX_ITEM_MESSAGE_TO_FACTOR, CX_ITEM_MESSAGE_TO_VARIABLE, CX_ITEM_INDIVIDUAL_MARGINAL (for dim 64 a marginal is
 * computed from the stored messages when cx_get_marginals reads it: the item only checks its variable).  A result with an undefined
 * dependency is not stored (the signal was not pending, src/signal.jl:668-730).
 * cx_update_batch is complete at return.  cx_update_batch_async is the same call without the wait for batches of up to 48 items
 * (dim 1 - 4): their records travel in the kernel arguments and the call returns when the launch is queued, like cx_sweep — what
 * a scheduler does next (set_value!, readiness bits: signal.jl:232-253) does not read the device, and every cx_get_* /
 * cx_residual / cx_sync waits for the stream.  Larger batches and dim 64 stage their records through a buffer of the handle and
 * are complete at return in both forms. */
int32_t cx_update_batch(cx_handle *h, const cx_item *items, int64_t n);
int32_t cx_update_batch_async(cx_handle *h, const cx_item *items, int64_t n);
/* values of the intermediates cx_update_batch keeps on the device: ProductOfMessages nodes (2 doubles each — dim > 1: a message
 * payload of d + d * d doubles each; the reference's default resolver creates them for variables of degree > 5; any degree — `form` CX_FORM_MOMENT or CX_FORM_NATURAL) and JointMarginal nodes (dim 1) (6 doubles each: mean[2], covariance[4] row-major, variables in
 * ascending id order).  A node never computed reads as NaN (UndefValue()). */
int32_t cx_get_products(cx_handle *h, int64_t n, const int64_t *variable_ids, const int32_t *range_lo, const int32_t *range_hi,
                        int32_t form, double *out);
int32_t cx_get_joint_marginals(cx_handle *h, int64_t n, const int64_t *factor_ids, double *out);
/* n_sweeps passes of the configured device schedule over the whole graph (asynchronous) */
int32_t cx_sweep(cx_handle *h, int32_t n_sweeps);
/* Damping for the schedules that iterate to a fixed point (CX_SCHED_FUSED, CX_SCHED_FLOODING; dim 1 - 4): every factor→variable message a
 * sweep computes is stored as (1 - lambda) x rule + lambda x the message it replaces, in natural parameters (an undefined old message does
 * not damp).  0 <= lambda < 1, 0 (the default) = off: the sweeps are then bit for bit what they were.  The fixed point is unchanged; loopy
 * Gaussian BP that oscillates outside the walk-summable regime converges under enough damping.  The reference has no counterpart: its rules
 * are the user's code, a user damps inside them.  Not with per-sweep message halos (cx_halo_configure); state halos are damped like the
 * plain sweeps they run. */
int32_t cx_set_damping(cx_handle *h, double lambda);
/* CX_SCHED_REFERENCE: ONE update_marginals!(engine, variable_ids) (src/inference_engine.jl:559-632; request_inference_for :298-323) for
 * the named variables in the caller's order — only what is pending for those marginals is computed (cx_sweep requests every variable
 * that is neither observed nor a stand-in, in ascending id order).  Other schedules: CX_ERR_UNSUPPORTED (they compute every message). */
int32_t cx_sweep_for(cx_handle *h, int64_t n, const int64_t *variable_ids);
/* CX_SCHED_REFERENCE: a user resolver's dependency wiring in place of DefaultDependencyResolver's (src/dependencies.jl:1-15): the i-th triple is
 * add_dependency!(signals[i], dependencies[i]; weak, intermediate, listen) (src/signal.jl:286-337), in call order (= dependency order per
 * signal).  Signals are named like batch items: CX_ITEM_MESSAGE_TO_FACTOR / CX_ITEM_MESSAGE_TO_VARIABLE (variable_id, factor_id),
 * CX_ITEM_INDIVIDUAL_MARGINAL (variable_id), CX_ITEM_JOINT_MARGINAL (factor_id).  The scheduler runs on whatever is wired; the VALUES come
 * from the device's rules, so a dependency list must be one a rule can serve:
 *   - a MessageToFactor / IndividualMarginal depends on MessageToVariable signals of its own variable: its value is the product of exactly
 *     its dependency list, in list order (a dropped dependency is left out of the product);
 *   - a MessageToVariable of a sum-product factor (additive, linear, linear-N) depends on MessageToFactor signals of its factor's other
 *     variables: its value is the factor's rule on the stored messages of the factor's other edges;
 *   - the messages of a CX_FACTOR_NORMAL_PRECISION factor (out ~ N(in, 1 / precision), precision Gamma-distributed) depend on MARGINALS — the
 *     variational rules of the reference's test processors (test/inference_engine_tests.jl:647-689, 939-1030), chosen by the dependency list as
 *     those rules choose:   to out / in     <- marginal(in / out), marginal(precision)              N(E[other], E[precision])
 *                           to precision    <- marginal(out), marginal(in)                          Gamma(3/2, 2 / (var out + var in + (E out - E in)^2))
 *                           to out / in     <- MessageToFactor(in / out), marginal(precision)       N(mean m, 1 / (var m + 1 / E[precision]))
 *                           JointMarginal   <- MessageToFactor(out), MessageToFactor(in), marginal(precision)    the 2-d Gaussian of :939-967
 *                           to precision    <- JointMarginal(factor)                                Gamma(3/2, 2 / (V11 - 2 V12 + V22 + (m1 - m2)^2))
 *     Such factors exist under this schedule only.  A variable on a CX_ROLE_PRECISION edge is Gamma-distributed everywhere: its messages are the
 *     natural pairs (shape - 1, rate) (CX_FORM_NATURAL; a prior is an opaque factor's message), its marginal is kept and returned as
 *     (shape, scale); Normal variables as (mean, variance).  Initial marginals and data: cx_set_marginals (CX_FORM_GAMMA; CX_FORM_MOMENT /
 *     CX_FORM_MEAN_PRECISION; CX_FORM_POINT = an observed value, variance 0), the user's set_value! on the marginal signal.
 * Anything else is CX_ERR_UNSUPPORTED, as are intermediate flags that close a cycle (process_dependencies! would never return) and a
 * dependency listed twice.  Two flags stand for calls other than add_dependency!:
 *   CX_WIRE_DEFAULT_VARIABLE  signals[i] = IndividualMarginal(v), dependencies[i] ignored: resolve_variable_dependencies!(
 *                             DefaultDependencyResolver(), engine, v) at this point of the call order (src/dependencies.jl:33-126: all-pairs up to
 *                             degree 5, a segment tree of ProductOfMessages above; a MessageToFactor is wired only if something listens to it by now)
 *   CX_WIRE_LINK              signals[i] = JointMarginal(f), dependencies[i] = IndividualMarginal(v): link_signal_to_variable!(v, signal)
 *                             (src/model_engine.jl:64-71): requested and computed with v's marginal (src/inference_engine.jl:313-315, 617-625)
 * Replaces the whole wiring (n == 0: none); allowed until the first value is set or the first call runs, as the reference wires at engine
 * construction.  The two fused families (cx_update_marginals) remain the fast path for the reference's two test models; this entry point runs
 * them, and any other wiring of these rules on any graph, call by call as the reference's scheduler would — mixed requests included.
 * dim 1: all of the above.  dim 2, 3, 4 (ABI 4): wirings of the sum-product rules (a MessageToFactor / marginal is the sum of ITS dependency
 * list); dim 64 forms marginals from all stored messages when they are read and refuses wirings. */
#define CX_WIRE_WEAK 1          /* add_dependency!(...; weak = true): the dependency need only be computed, not fresh (src/signal.jl:36-45) */
#define CX_WIRE_INTERMEDIATE 2  /* intermediate = true: process_dependencies! descends through it (src/signal.jl:466-490) */
#define CX_WIRE_NO_LISTEN 4     /* listen = false: the dependency's set_value! does not make the signal potentially pending */
#define CX_WIRE_DEFAULT_VARIABLE 8
#define CX_WIRE_LINK 16
int32_t cx_graph_wire(cx_handle *h, int64_t n, const cx_item *signals, const cx_item *dependencies, const int32_t *flags);
/* the plan the last reference-order call replayed: out8 = { stages, kernel launches, executions (signals computed), of which messages,
 * passes of the reference's loop (the final marginal round included), plans kept, calls that replayed a kept plan, calls that had to
 * run the scheduler }.  Zeros before the first call and for other schedules. */
int32_t cx_ref_plan_stats(const cx_handle *h, int64_t *out8);
/* the scalar chain scan as ONE launch (CX_SCHED_CHAIN_SCAN, the scans of CX_SCHED_TREE's heavy paths, the variational families' state
 * pass): a single-pass scan whose workgroups publish their tile totals under the launch's tag instead of ending a kernel
 * (csrc/cx_chain.hip: k_chain_onepass).  Taken when the whole grid is resident at once (up to twice the compute units' count of tiles); every wait is bounded in time (0.5 s, CX_CHAIN_ONEPASS_TIMEOUT_MS) —
 * a launch whose wait times out stores NOTHING, the next call that checks the device returns CX_ERR_DEVICE, the handle goes back to the
 * two-launch scan and the caller repeats the sweep (a chain-scan sweep is exact whatever it starts from).  CX_CHAIN_ONEPASS=0 turns it off.
 * out4 = { 1 ready / 0 not prepared / -1 off, launches of the one-launch form so far, G, 0 } — G counts another kind of fused launch: the
 * sweeps between two exchanges of a deep-halo partition (cx_halo_configure_state + cx_halo_set_layers) replayed as ONE graph launch once
 * the same batch has been asked for twice — with CX_HALO_GRAPH=1 only: measured 2 - 5 % slower than the plain launches on MI355X, off by default. */
int32_t cx_chain_scan_stats(const cx_handle *h, int64_t *out4);
/* the XCD-resident cluster (reference-order plans of many dependent stages of 1 - 16 k items — calls on loopy graphs — run as ONE launch
 * of the workgroups of one XCD behind barriers that stay in that XCD's L2): out4 = { 1 ready / 0 not prepared / -1 off
 * (CX_REF_CLUSTER=0, a device that is neither gfx942 nor gfx950, or a barrier once timed out), workgroups per launch (compute units),
 * calls RECOVERED (every wait of the cluster is bounded in time — 1.5 s, CX_REF_CLUSTER_TIMEOUT_MS; a call whose cluster gives up is
 * finished from its first incomplete stage on plain launches, returns CX_OK with the same results, and the handle stays with launches),
 * 1 when the last reference-order call ran on the cluster }. */
int32_t cx_cluster_stats(const cx_handle *h, int64_t *out4);
/* the executions of the last reference-order call in the reference's order — what a `trace = true` engine records
 * (src/inference_engine.jl:650-862: TracedInferenceExecution.signal), as items (kind, variable_id, factor_id | CX_ITEM_RANGE).
 * *n_executions = their number; out (may be NULL) receives the first `capacity` of them. */
int32_t cx_ref_trace(const cx_handle *h, int64_t capacity, cx_item *out, int64_t *n_executions);
/* max over directed messages of |Δmean|, |Δvariance| between the last two sweeps (host-synchronous) */
int32_t cx_residual(cx_handle *h, double *out_max_abs_delta);
/* sweep until cx_residual over `check_every` sweeps is <= tol, or max_sweeps have run (stopping rule for loopy graphs; the
 * reference leaves it to the caller of update_marginals!).  A NaN residual (undefined messages) never satisfies tol. */
int32_t cx_sweep_until(cx_handle *h, double tol, int32_t max_sweeps, int32_t check_every, int32_t *sweeps_run, double *residual);

/* ---- partitioned graphs (one handle per GPU; exchange is the caller's: RCCL/torch.distributed) --
 * Cut edges appear in this rank's graph as degree-1 "ghost" variables whose variable→factor message is
 * produced by another rank.  cx_halo_configure names the edges this rank exports / imports; buffers hold
 * NATURAL-form payloads (2 doubles per edge for dim == 1), in list order. */
int32_t cx_halo_configure(cx_handle *h, int64_t n_send, const int64_t *send_variable_ids,
                          const int64_t *send_factor_ids, int64_t n_recv, const int64_t *recv_variable_ids,
                          const int64_t *recv_factor_ids);
int32_t cx_halo_buffers(cx_handle *h, void **send_device_ptr, int64_t *send_bytes, void **recv_device_ptr,
                        int64_t *recv_bytes);
/* use caller-owned device buffers (e.g. torch tensors handed to torch.distributed) instead of the library's */
int32_t cx_halo_set_buffers(cx_handle *h, void *send_device_ptr, void *recv_device_ptr);
/* one partitioned sweep = three asynchronous calls, so the caller can overlap the exchange with the main kernel:
 *   cx_sweep_begin : compute the exported variable→factor messages and pack them into the send buffer
 *   [caller starts the exchange of send → peer's recv buffer on its communication stream]
 *   cx_sweep_main  : the sweep over every variable this rank owns
 *   [caller makes the handle's stream wait for the exchange]
 *   cx_sweep_end   : unpack the recv buffer into the ghost variables and push it through the cut factors
 * The three calls together compute exactly one cx_sweep of the un-partitioned graph. */
int32_t cx_sweep_begin(cx_handle *h);
int32_t cx_sweep_main(cx_handle *h);
int32_t cx_sweep_end(cx_handle *h);

/* The same three steps with the exchange issued by the library itself on RCCL (grouped ncclSend/ncclRecv with the
 * partition neighbours on a dedicated stream, overlapped with the main kernel) — no host language in the per-sweep loop.
 *   cx_comm_unique_id : rank 0 obtains the 128-byte ncclUniqueId and distributes it by any means
 *   cx_comm_init      : every rank joins (one communicator per handle)
 *   cx_halo_peers     : which segments (in messages) of the send / recv lists of cx_halo_configure belong to which rank
 *   cx_sweep_exchange : n_sweeps x { begin, exchange, main, end }, asynchronous on the handle's stream */
int32_t cx_comm_unique_id(void *out128);
int32_t cx_comm_init(cx_handle *h, int32_t world, int32_t rank, const void *id128);
int32_t cx_halo_peers(cx_handle *h, int32_t n_peers, const int32_t *peer_rank, const int64_t *send_offset,
                      const int64_t *send_count, const int64_t *recv_offset, const int64_t *recv_count);
int32_t cx_sweep_exchange(cx_handle *h, int32_t n_sweeps);

/* ---- state halos ("deep halo"): exchange once per `depth` sweeps instead of once per sweep ---------------------------
 * The local graph of a rank additionally holds `depth` layers of its neighbours' variables (redundant copies, with all
 * their factors; beyond them a degree-1 stand-in per cut factor).  Between exchanges the handle runs plain cx_sweep calls;
 * an exchange overwrites every factor→variable message of the redundant variables with the values their owner holds.
 * After k <= depth sweeps every message of an OWNED variable equals the un-partitioned sweep's bit for bit: the error of
 * the frozen outer edge advances one layer per sweep and is wiped by the next exchange.  The redundant work is
 * 2*depth layers per rank; the per-sweep cost of the halo falls by the factor depth.
 *   cx_halo_configure_state : (variable, factor) of the messages exported to / imported from the neighbours, grouped by
 *                             peer like cx_halo_configure (cx_halo_peers, cx_halo_buffers, cx_halo_set_buffers apply).  Any dim:
 *                             a message travels in its storage form — 2 doubles (dim 1), d + d(d+1)/2 (dim 2..4: natural
 *                             parameters, packed symmetric), 64 + 64*64 (dim 64)
 *   cx_halo_state_pack / _unpack : gather into the send buffer / scatter from the recv buffer (caller-owned transport)
 *   cx_halo_state_exchange  : pack, grouped ncclSend/ncclRecv, unpack — all on the handle's stream (cx_comm_init first) */
int32_t cx_halo_configure_state(cx_handle *h, int64_t n_send, const int64_t *send_var, const int64_t *send_fac,
                                int64_t n_recv, const int64_t *recv_var, const int64_t *recv_fac);
/*   cx_halo_set_layers      : optional.  layer[i] = distance (1 .. depth; stand-ins depth + 1) of redundant variable_ids[i] from the
 *                             owned set.  cx_sweep then runs, in the j-th sweep after an exchange, only the slices that hold
 *                             variables of layer <= depth - j + 1 — the layers that can still be valid — instead of the whole local
 *                             graph (fused schedule).  Owned results are unchanged bit for bit. */
int32_t cx_halo_set_layers(cx_handle *h, int64_t n, const int64_t *variable_ids, const int32_t *layer, int32_t depth);
int32_t cx_halo_state_pack(cx_handle *h);
int32_t cx_halo_state_unpack(cx_handle *h);
int32_t cx_halo_state_exchange(cx_handle *h);
/*   cx_halo_exchange_sweep  : one batch = the exchange AND n_sweeps sweeps, the exchange overlapped with the first sweep: the slices
 *                             that hold owned variables only (known from cx_halo_set_layers) run on the handle's stream while a
 *                             second stream packs, sends / receives and unpacks; the redundant rows' part of that sweep follows the
 *                             unpack.  Bit-identical to cx_halo_state_exchange + cx_sweep(n_sweeps), and falls back to exactly that
 *                             when the handle cannot split a sweep (no layers, dim > 1, another schedule). */
int32_t cx_halo_exchange_sweep(cx_handle *h, int32_t n_sweeps);
/* The same exchange WITHOUT a collective library (dim 1 - 4): every rank pushes its boundary state straight into a receive area of
 * the neighbour — device memory the neighbour exported with hipIpcGetMemHandle — and raises an epoch flag there; the neighbour's
 * unpack kernel waits for the flag.  Two launches on the handle's stream per exchange (push; wait + unpack) instead of pack,
 * RCCL kernel, unpack.  Results are bit-identical to cx_halo_state_exchange.
 *   cx_halo_ipc_alloc     : after cx_halo_configure_state + cx_halo_peers.  Allocates this rank's block (flags + two receive areas)
 *                           and returns its 64-byte hipIpcMemHandle_t, its device address (for a neighbour in the SAME process,
 *                           which cannot open the handle) and the size of one receive area.  The caller carries the three to the
 *                           neighbours (torch.distributed.all_gather_object, MPI, a file).
 *   cx_halo_ipc_connect   : peer entry peer_index (order of cx_halo_peers) sends to the neighbour's block: handle64 XOR
 *                           same_process_base; remote_entry = the neighbour's peer entry that receives from this rank,
 *                           remote_recv_off = that entry's recv_offset (messages), remote_area_bytes = the neighbour's area size.
 *   cx_halo_ipc_push      : store the boundary state into the neighbours' receive areas of the next epoch, raise their flags
 *   cx_halo_ipc_unpack    : wait for this rank's flags of the epoch last pushed, scatter the receive area into the redundant rows
 *   cx_halo_ipc_exchange  : push, then unpack; asynchronous like cx_sweep.  Every neighbour has to call it the same number of
 *                           times.  A neighbour that does not arrive within the timeout (default 20 s) is NOT waited for for
 *                           ever: the unpack gives up, the grid drains and cx_halo_ipc_status reports it.
 *   cx_halo_ipc_status    : synchronises; *timed_out != 0 when an unpack gave up (results since are void), *exchanges = count */
int32_t cx_halo_ipc_alloc(cx_handle *h, void *handle64, void **local_base, int64_t *area_bytes);
int32_t cx_halo_ipc_connect(cx_handle *h, int32_t peer_index, const void *handle64, void *same_process_base, int32_t remote_entry,
                            int64_t remote_recv_off, int64_t remote_area_bytes);
int32_t cx_halo_ipc_push(cx_handle *h);
int32_t cx_halo_ipc_unpack(cx_handle *h);
int32_t cx_halo_ipc_exchange(cx_handle *h);
/*   cx_halo_ipc_exchange_sweep : one batch = push | the owned part of the first sweep | wait + unpack | the rest of that sweep |
 *                           sweeps 2 .. n, all on the handle's stream: the neighbours' pushes travel while this rank computes
 *                           its interior.  Bit-identical to cx_halo_ipc_exchange + cx_sweep(n), and exactly that when the sweep
 *                           cannot be split (no cx_halo_set_layers, dim > 1, another schedule). */
int32_t cx_halo_ipc_exchange_sweep(cx_handle *h, int32_t n_sweeps);
/*   cx_halo_ipc_batch     : one batch of n sweeps (2 <= n <= depth) with the push of the NEXT exchange inside its last sweep: that
 *                           sweep runs the slices that write the boundary state first, pushes, then runs the rest; the batch's own
 *                           exchange is unpacked after the owned part of its first sweep.  Between a push and the wait for it then
 *                           lie two partial sweeps of compute for the transfer to hide behind.  Bit-identical to
 *                           cx_halo_ipc_exchange + cx_sweep(n); afterwards the next exchange is pushed but not unpacked, which
 *                           cx_halo_ipc_exchange / _exchange_sweep / _batch pick up.  Falls back to cx_halo_ipc_exchange_sweep.
 *   cx_halo_ipc_set_fused : on != 0 — the caller asserts that every neighbour pushes from ANOTHER device: push and unpack of an
 *                           exchange may then share one launch.  Default off (two launches): a rank that is its own neighbour,
 *                           handles of one process and processes sharing a GPU must not wait inside a kernel for workgroups of the
 *                           same or a co-scheduled kernel. */
int32_t cx_halo_ipc_batch(cx_handle *h, int32_t n_sweeps);
int32_t cx_halo_ipc_set_fused(cx_handle *h, int32_t on);
int32_t cx_halo_ipc_status(cx_handle *h, int32_t *timed_out, int64_t *exchanges);
int32_t cx_halo_ipc_set_timeout(cx_handle *h, double seconds);

/* ---- partitioned chain scan (CX_SCHED_CHAIN_SCAN; SURVEY.md §8e: "contiguous time blocks + one composed map per block") ----
 * A rank holds a time block of a chain (its own variables, the cut transition factors, the remote end of each as a degree-1
 * stand-in named in cx_halo_configure).  The exact forward / backward messages of a block are a projective-linear function of
 * the ONE message that enters it at either end, so the ranks exchange maps, not sweeps:
 *   cx_chain_block_maps : the block's composed forward and backward maps over its links — 6 doubles each, (e f g A B C) of
 *                         [xi' w' 1] ~ [[e f g] [0 A B] [0 C 1]] [xi w 1] — and the side sums (all non-chain messages) of its
 *                         first and last variable, from the data currently on the device.  Before the call the caller sets the
 *                         factor→variable messages of the cut factors into the block's end variables to natural (0, 0), so that
 *                         the maps exclude them.
 *   [caller all-gathers the maps (16 doubles per rank), composes the prefixes and sets the stand-ins' variable→factor messages]
 *   cx_sweep(h, 1)      : the block's exact messages and marginals (cortex.jl_amd/partition.py: ChainScanExchange).
 * dim 2, 3, 4 (round 3): the same protocol with the maps of csrc/cx_mvchain.hip.  A map is ND = 2 d(d+1)/2 + d^2 + 2 d doubles —
 *   P (packed upper) | B (row-major) | C (packed upper) | h | c  of  f(eta, Lambda) = (c + B (Lambda + P)^-1 (eta + h), C - B (Lambda + P)^-1 B')
 *   — in forward6 / backward6 (ND doubles each); the side sums are eta[d] | Lambda (packed upper) in side_first2 / side_last2.  The
 *   stand-ins of a dim > 1 block are named with cx_halo_configure (recv lists only matter: they mark the stand-in variables).
 * dim 64 (round 4): the same layout (ND = 8,384, sides 2,144 doubles).  The block's composition tree (csrc/cx_mv64chain.hip) ends in
 *   ONE potential of its two end variables; the call runs the compose launches and returns it as the two directions' maps.  dim 64
 *   keeps one message buffer and no variable→factor messages, so the caller hands the boundary over AFTER the cut factor's rule: as
 *   the cut factor's factor→variable message into the block's end variable (cx_set_messages, CX_TO_VARIABLE).  A cx_sweep that
 *   follows with nothing but those two messages changed starts from the potentials that are on the device already (its walks only). */
int32_t cx_chain_block_maps(cx_handle *h, double *forward6, double *backward6, double *side_first2, double *side_last2,
                            int64_t *first_variable_id, int64_t *last_variable_id, int64_t *n_links);

/* ---- the chain-scan schedule for dim 64 (csrc/cx_mv64chain.hip) ----
 * One cx_sweep composes pairwise potentials of blocks of links bottom-up and walks them top-down (the plan: csrc/cx_chain64_plan.h),
 * so that ONE call is the reference's one update_marginals! on a chain (src/inference_engine.jl:575-608).  cx_chain_plan_stats
 * reports the plan the last cx_sweep used, for the work count of a measurement: out8 = {links per level-0 block, potentials per
 * group, levels, potentials, pairwise compositions per sweep (960 matrix instructions each), rule applications per sweep (384
 * each), kernel launches per sweep, device bytes of the plan}.  All zeros before the first sweep and for any other handle. */
int32_t cx_chain_plan_stats(const cx_handle *h, int64_t *out8);

/* CX_SCHED_TREE: the plan of the last sweep (zeros before the first one and for other schedules).
 * out8 = { depth, stages, items, k-ary entries, components, messages upwards, messages downwards, marginals }. */
int32_t cx_tree_plan_stats(const cx_handle *h, int64_t *out8);
/* CX_SCHED_TREE: when the sweep takes fewer launches that way, the plan runs over HEAVY PATHS — every variable's
 * heaviest child continues its path, through a two-edge factor or through a factor of more edges (which, given the messages of its
 * other variables, is a pairwise rule between the two: formed on the device before the depth's first scan); the paths of one light
 * depth (light edges above them) are ONE segmented scan per direction, whatever their length, the light edges stay items: O(log n)
 * rounds of launches instead of 2 x depth + 1 (a chain of T states with a latent layer below each: ~ 25 launches instead of ~ 2 T; a
 * tree of 1.1 M edges and 144,559 levels: 32).  The same messages, every marginal exact.
 * out4 = { light depths, paths of two or more variables, variables on no such path, launches per sweep }; zeros when the level
 * schedule is in use (then cx_tree_plan_stats's "stages" are its launches).  With heavy paths "stages" / "items" count the item stages. */
int32_t cx_tree_heavy_path_stats(const cx_handle *h, int64_t *out4);

/* ---- numerical guards as counters (non-finite values, variances <= 0 and matrices that are not positive definite are
 * reported through status codes and counters, never by an abort; the reference has nothing to mirror here — a rule that divides by
 * zero throws in the user's Julia code).  An undefined value is NaN and propagates by itself, a rule whose input is not positive
 * definite leaves its output undefined or unchanged; this call counts, on the device, the stored factor→variable messages INTO
 * NON-OBSERVED variables:  out4 = { defined, undefined (UndefValue: never computed, or a dependency was undefined), defined with a
 * negative precision (dim > 1: a negative diagonal entry of the precision matrix), defined with a non-finite entry (a point mass
 * (y, +inf) of dim 1 is a value, not counted) }.  Synchronous. */
int32_t cx_message_health(cx_handle *h, int64_t *out4);

/* ---- checkpoint (the reference keeps no persistent state — src/ has no serialisation at all) ----
 * The mutable state of a handle (every message buffer, the marginals, the observed-variable flags, the sweep counter)
 * as one relocatable host blob.  A blob restores only into a handle created with the same dim / family / schedule and
 * the same graph AND the same rule parameters (a fingerprint of the flattened graph, the factor parameters and the (A, Q) sets is
 * checked: a blob continues under the parameters it was exported with, or not at all; blobs of an earlier format are refused with
 * a version error); after cx_state_import the handle continues exactly
 * where the exporting one stood: the following sweeps reproduce its results bit for bit.  Halo buffers are not part
 * of the state (the next partitioned sweep exchanges them again).  Variational handles export their marginals and observed
 * flags (the structured family also the inner chain handle's state, inside the same blob). */
int32_t cx_state_bytes(const cx_handle *h, int64_t *bytes);
int32_t cx_state_export(cx_handle *h, void *buf, int64_t bytes);        /* synchronises the handle's stream */
int32_t cx_state_import(cx_handle *h, const void *buf, int64_t bytes);  /* validates everything before writing */

/* ---- measurement ------------------------------------------------------------------------------ */
#define CX_KERNEL_VAR_TO_FACTOR 0
#define CX_KERNEL_FACTOR_TO_VAR 1
#define CX_KERNEL_FUSED 2
#define CX_KERNEL_BATCH 3
#define CX_KERNEL_BIG_VAR 4
#define CX_KERNEL_HALO_BEGIN 5
#define CX_KERNEL_HALO_END 6
/* 7: unused since ABI 4 (was the two-sweeps-per-launch kernel) */
#define CX_KERNEL_COUNT 8
/* hipEvent pairs around kernel launches on the handle's stream: on == 1 every launch, on == n > 1 every n-th
 * launch of each kernel (events are barrier packets; a stride keeps the other launches back to back), 0 off */
int32_t cx_profile_enable(cx_handle *h, int32_t on);
int32_t cx_profile_read(cx_handle *h, int32_t kernel, double *total_ms, int64_t *launches); /* syncs; resets */
const char *cx_kernel_name(int32_t kernel);

#ifdef __cplusplus
}
#endif
#endif /* CORTEX_HIP_H */
