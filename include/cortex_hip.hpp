// cortex_hip.hpp — C++17 host classes over the C ABI of cortex_hip.h (header-only; nothing here is part of the ABI).
//
// SURVEY.md §8b asks for "a Julia subtype (shim) and an equivalent C++ host class, both owning a cx_handle*".  The Julia
// shim is julia/CortexHIP.jl; this is the C++ side:
//
//   cortex::Handle        RAII owner of a cx_handle; every method is one ABI call, a non-zero status becomes cortex::Error
//                         carrying cx_last_error() (the analogue of the shim's `error(unsafe_string(cx_last_error(h)))`).
//                         Exceptions exist only on THIS side of the boundary — none crosses the C ABI.
//   cortex::HipProcessor  the plugin of src/inference_engine.jl:331-509 in the two forms a host scheduler drives:
//                           process(kind, variable, factor)  = `process!` overridden to ENQUEUE (precedent: the
//                                                              InferenceRequestScanner, inference_engine.jl:528-537);
//                           flush()                          = one cx_update_batch for everything enqueued, in order;
//                           update_marginals(ids, sweeps)    = `update_marginals!` specialised on the processor type:
//                                                              cx_sweep + cx_get_marginals.
//   cortex::VmpProcessor  the variational families (weak dependencies): set_marginal / update_marginals / marginals.
//
// tests/cpp/host_class_demo.cpp drives both against the reference's SSM test graphs; tests/test_gpu_cpp_host.py checks
// its output against the exact smoother and the array form of the variational updates.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <array>
#include <vector>

#include "cortex_hip.h"

namespace cortex {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string &msg) : std::runtime_error("cortex_hip status " + std::to_string(c) + ": " + msg), code(c) {}
};

inline cx_config make_config(int32_t device = 0, int32_t dim = 1, int32_t schedule = CX_SCHED_FUSED, int32_t family = CX_FAMILY_GAUSSIAN,
                             bool marginals_in_sweep = true, bool materialize_messages_to_factor = false) {
    cx_config c{};
    c.struct_size = (int32_t)sizeof(cx_config);
    c.device = device; c.dim = dim; c.schedule = schedule; c.family = family;
    c.compute_marginals_in_sweep = marginals_in_sweep ? 1 : 0;
    c.materialize_messages_to_factor = materialize_messages_to_factor ? 1 : 0;
    return c;
}

class Handle {
  public:
    explicit Handle(const cx_config &cfg) : dim_(cfg.dim) {
        const int32_t rc = cx_create(&cfg, &h_);
        if (rc != CX_OK) throw Error(rc, cx_last_error(nullptr));
    }
    ~Handle() { if (h_) (void)cx_destroy(h_); }
    Handle(const Handle &) = delete;
    Handle &operator=(const Handle &) = delete;
    Handle(Handle &&o) noexcept : h_(o.h_), dim_(o.dim_) { o.h_ = nullptr; }
    Handle &operator=(Handle &&o) noexcept { if (this != &o) { if (h_) (void)cx_destroy(h_); h_ = o.h_; dim_ = o.dim_; o.h_ = nullptr; } return *this; }

    cx_handle *get() const { return h_; }
    int32_t dim() const { return dim_; }

    // graph ingestion: the flattened BipartiteFactorGraph (model_engine.jl:329-391)
    void graph_create(const std::vector<int64_t> &edge_var, const std::vector<int64_t> &edge_fac, const std::vector<int64_t> &factor_ids,
                      const std::vector<int32_t> &factor_kind, const std::vector<double> &factor_params /* CX_NPARAM per factor, or empty */,
                      const std::vector<int32_t> &edge_role = {}) {
        if (edge_var.size() != edge_fac.size() || factor_ids.size() != factor_kind.size() ||
            (!edge_role.empty() && edge_role.size() != edge_var.size()) ||
            (!factor_params.empty() && factor_params.size() != factor_ids.size() * CX_NPARAM))
            throw Error(CX_ERR_INVALID_ARGUMENT, "graph_create: array lengths disagree");
        std::vector<double> zeros;
        const double *params = factor_params.data();
        if (factor_params.empty()) { zeros.assign(factor_ids.size() * CX_NPARAM, 0.0); params = zeros.data(); }
        check(cx_graph_create(h_, (int64_t)edge_var.size(), edge_var.data(), edge_fac.data(), edge_role.empty() ? nullptr : edge_role.data(),
                              (int64_t)factor_ids.size(), factor_ids.data(), factor_kind.data(), params));
    }
    void set_factor_matrices(int64_t parameter_set, const std::vector<double> &A, const std::vector<double> &Q) {
        check(cx_set_factor_matrices(h_, parameter_set, A.data(), Q.data()));
    }
    cx_stats stats() const { cx_stats s{}; check(cx_graph_stats(h_, &s)); return s; }
    // CX_SCHED_TREE: { depth, stages, items, k-ary entries, components, messages up, messages down, marginals } of the last sweep's plan
    std::array<int64_t, 8> tree_plan_stats() const { std::array<int64_t, 8> o{}; check(cx_tree_plan_stats(h_, o.data())); return o; }
    std::array<int64_t, 4> tree_heavy_path_stats() const { std::array<int64_t, 4> o{}; check(cx_tree_heavy_path_stats(h_, o.data())); return o; }

    // data injection / read-back: the user's set_value! on message signals (signal.jl:232-253)
    void set_messages(const std::vector<int64_t> &variable_ids, const std::vector<int64_t> &factor_ids, int32_t direction, int32_t form,
                      const std::vector<double> &payload) {
        need(payload.size(), variable_ids.size() * (size_t)cx_payload_doubles(dim_, form), "set_messages payload");
        check(cx_set_messages(h_, (int64_t)variable_ids.size(), variable_ids.data(), factor_ids.data(), direction, form, payload.data()));
    }
    std::vector<double> get_messages(const std::vector<int64_t> &variable_ids, const std::vector<int64_t> &factor_ids, int32_t direction,
                                     int32_t form = CX_FORM_MOMENT) const {
        std::vector<double> out(variable_ids.size() * (size_t)cx_payload_doubles(dim_, CX_FORM_MOMENT));
        check(cx_get_messages(h_, (int64_t)variable_ids.size(), variable_ids.data(), factor_ids.data(), direction, form, out.data()));
        return out;
    }
    void seed_messages(int32_t direction, double mean, double variance) { check(cx_seed_messages(h_, direction, mean, variance)); }
    std::vector<double> get_marginals(const std::vector<int64_t> &variable_ids) const {
        std::vector<double> out(variable_ids.size() * (size_t)cx_payload_doubles(dim_, CX_FORM_MOMENT));
        check(cx_get_marginals(h_, (int64_t)variable_ids.size(), variable_ids.data(), out.data()));
        return out;
    }

    // compute
    void update_batch(const std::vector<cx_item> &items) { check(cx_update_batch(h_, items.data(), (int64_t)items.size())); }
    void update_batch_async(const std::vector<cx_item> &items) { check(cx_update_batch_async(h_, items.data(), (int64_t)items.size())); }
    void sweep(int32_t n = 1) { check(cx_sweep(h_, n)); }
    // CX_SCHED_REFERENCE: ONE update_marginals!(engine, variable_ids) — the named variables in the caller's order, only what is pending for them
    void sweep_for(const std::vector<int64_t> &variable_ids) { check(cx_sweep_for(h_, (int64_t)variable_ids.size(), variable_ids.data())); }
    std::array<int64_t, 4> chain_scan_stats() const { std::array<int64_t, 4> o{}; check(cx_chain_scan_stats(h_, o.data())); return o; }      // state, one-launch scans so far, 0, 0
    std::array<int64_t, 8> ref_plan_stats() const { std::array<int64_t, 8> o{}; check(cx_ref_plan_stats(h_, o.data())); return o; }      // stages, launches, executions, messages, passes, plans, hits, misses
    std::vector<cx_item> ref_trace() const {       // the executions of the last reference-order call, in the reference's order
        int64_t n = 0;
        check(cx_ref_trace(h_, 0, nullptr, &n));
        std::vector<cx_item> out((size_t)n);
        if (n) check(cx_ref_trace(h_, n, out.data(), &n));
        return out;
    }
    // a user resolver's wiring: one triple per add_dependency!(signal, dependency; flags) (CX_WIRE_*), right after graph_create
    void graph_wire(const std::vector<cx_item> &signals, const std::vector<cx_item> &dependencies, const std::vector<int32_t> &flags) {
        need(dependencies.size(), signals.size(), "graph_wire dependencies"); need(flags.size(), signals.size(), "graph_wire flags");
        check(cx_graph_wire(h_, (int64_t)signals.size(), signals.data(), dependencies.data(), flags.data()));
    }
    void set_damping(double lambda) { check(cx_set_damping(h_, lambda)); }      // fused / flooding sweeps: new = (1 - lambda) rule + lambda old
    double residual() { double r = 0; check(cx_residual(h_, &r)); return r; }
    std::array<int64_t, 4> message_health() { std::array<int64_t, 4> o{}; check(cx_message_health(h_, o.data())); return o; }      // defined, undefined, negative precision, non-finite
    std::pair<int32_t, double> sweep_until(double tol, int32_t max_sweeps, int32_t check_every = 10) {
        int32_t n = 0; double r = 0;
        check(cx_sweep_until(h_, tol, max_sweeps, check_every, &n, &r));
        return {n, r};
    }
    void sync() { check(cx_sync(h_)); }
    void set_stream(void *hip_stream) { check(cx_set_stream(h_, hip_stream)); }

    // variational families
    void set_marginals(const std::vector<int64_t> &variable_ids, int32_t form, const std::vector<double> &payload) {
        need(payload.size(), variable_ids.size() * (form == CX_FORM_POINT ? 1u : 2u), "set_marginals payload");
        check(cx_set_marginals(h_, (int64_t)variable_ids.size(), variable_ids.data(), form, payload.data()));
    }
    void update_marginals(const std::vector<int64_t> &variable_ids) { check(cx_update_marginals(h_, (int64_t)variable_ids.size(), variable_ids.data())); }
    void update_marginals_of_class(int64_t which /* CX_VMP_ALL_NORMAL | CX_VMP_ALL_PRECISION */) { check(cx_update_marginals(h_, which, nullptr)); }

    // checkpoint
    std::vector<unsigned char> export_state() {
        int64_t n = 0;
        check(cx_state_bytes(h_, &n));
        std::vector<unsigned char> blob((size_t)n);
        check(cx_state_export(h_, blob.data(), n));
        return blob;
    }
    void import_state(const std::vector<unsigned char> &blob) { check(cx_state_import(h_, blob.data(), (int64_t)blob.size())); }

    // partitions (one process per GPU): deep halo and per-sweep message halo
    void halo_configure_state(const std::vector<int64_t> &sv, const std::vector<int64_t> &sf, const std::vector<int64_t> &rv, const std::vector<int64_t> &rf) {
        check(cx_halo_configure_state(h_, (int64_t)sv.size(), sv.data(), sf.data(), (int64_t)rv.size(), rv.data(), rf.data()));
    }
    void halo_configure(const std::vector<int64_t> &sv, const std::vector<int64_t> &sf, const std::vector<int64_t> &rv, const std::vector<int64_t> &rf) {
        check(cx_halo_configure(h_, (int64_t)sv.size(), sv.data(), sf.data(), (int64_t)rv.size(), rv.data(), rf.data()));
    }
    void halo_peers(const std::vector<int32_t> &rank, const std::vector<int64_t> &send_off, const std::vector<int64_t> &send_count,
                    const std::vector<int64_t> &recv_off, const std::vector<int64_t> &recv_count) {
        check(cx_halo_peers(h_, (int32_t)rank.size(), rank.data(), send_off.data(), send_count.data(), recv_off.data(), recv_count.data()));
    }
    void comm_init(int32_t world, int32_t rank, const void *id128) { check(cx_comm_init(h_, world, rank, id128)); }
    void halo_state_exchange() { check(cx_halo_state_exchange(h_)); }
    void halo_exchange_sweep(int n_sweeps) { check(cx_halo_exchange_sweep(h_, n_sweeps)); }
    void halo_ipc_exchange() { check(cx_halo_ipc_exchange(h_)); }
    void halo_ipc_exchange_sweep(int n_sweeps) { check(cx_halo_ipc_exchange_sweep(h_, n_sweeps)); }
    void sweep_exchange(int32_t n = 1) { check(cx_sweep_exchange(h_, n)); }

  private:
    void check(int32_t rc) const { if (rc != CX_OK) throw Error(rc, cx_last_error(h_)); }
    static void need(size_t got, size_t want, const char *what) {
        if (got != want) throw Error(CX_ERR_INVALID_ARGUMENT, std::string(what) + ": " + std::to_string(got) + " doubles, expected " + std::to_string(want));
    }
    cx_handle *h_ = nullptr;
    int32_t dim_ = 1;
};

// The sum-product processor.  A host scheduler (the reference's update_marginals! loop, or any restatement of it) calls
// process() where the reference calls process!(processor, engine, variable_id, signal); signals that are mutually independent
// may be enqueued together and computed by one flush().
class HipProcessor {
  public:
    explicit HipProcessor(Handle &h) : h_(h) {}
    void process(int32_t kind /* CX_ITEM_* */, int64_t variable_id, int64_t factor_id = 0) {
        cx_item it{};
        it.kind = kind; it.variable_id = variable_id; it.factor_id = factor_id;
        queue_.push_back(it);
    }
    size_t pending() const { return queue_.size(); }
    void flush() {
        if (queue_.empty()) return;
        std::vector<cx_item> q;
        q.swap(queue_);          // the queue is empty again even if the batch fails
        h_.update_batch(q);
        launches_++;
    }
    // update_marginals!(engine, ids) as a whole-call override: n_sweeps passes of the device schedule, then the marginals
    std::vector<double> update_marginals(const std::vector<int64_t> &variable_ids, int32_t n_sweeps = 1) {
        flush();
        h_.sweep(n_sweeps);
        // the sweep writes the marginals of the messages it STARTED from; refresh the requested ones from what it produced
        for (int64_t v : variable_ids) process(CX_ITEM_INDIVIDUAL_MARGINAL, v);
        flush();
        return h_.get_marginals(variable_ids);
    }
    int64_t launches() const { return launches_; }

  private:
    Handle &h_;
    std::vector<cx_item> queue_;
    int64_t launches_ = 0;
};

// The variational families: the state is the set of marginals.
class VmpProcessor {
  public:
    explicit VmpProcessor(Handle &h) : h_(h) {}
    void observe(int64_t variable_id, double y) { h_.set_marginals({variable_id}, CX_FORM_POINT, {y}); }
    void set_normal(int64_t variable_id, double mean, double precision) { h_.set_marginals({variable_id}, CX_FORM_MEAN_PRECISION, {mean, precision}); }
    void set_gamma(int64_t variable_id, double shape, double scale) { h_.set_marginals({variable_id}, CX_FORM_GAMMA, {shape, scale}); }
    void update_marginals(const std::vector<int64_t> &variable_ids) { h_.update_marginals(variable_ids); }
    std::vector<double> marginals(const std::vector<int64_t> &variable_ids) const { return h_.get_marginals(variable_ids); }

  private:
    Handle &h_;
};

}  // namespace cortex
