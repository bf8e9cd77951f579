# julia/CortexHIP.jl — the reference-side binding of libcortex_hip.so (SURVEY.md §8f1).
#
# NOT EXECUTED in the authoring container (no Julia toolchain, no network): shipped as source for a maintainer to try
# against Cortex.jl v0.3.0.  It uses only the C ABI of include/cortex_hip.h; the identical call sequence is exercised
# through Python ctypes by every `-m gpu` test of this repository (cortex.jl_amd/hip_processor.py is the tested twin).
# The HipProcessor part is the same text as the stub in INTEGRATION.md; the variational, checkpoint and multi-GPU parts
# follow the entry-point table there.

module CortexHIP
using Cortex
const lib = "libcortex_hip.so"

struct CxConfig            # mirrors cx_config
    struct_size::Int32; device::Int32; dim::Int32; schedule::Int32
    compute_marginals_in_sweep::Int32; materialize_messages_to_factor::Int32; family::Int32; reserved::Int32
end
struct CxItem              # mirrors cx_item
    kind::Int32; reserved::Int32; variable_id::Int64; factor_id::Int64
end

mutable struct HipProcessor <: Cortex.AbstractInferenceRequestProcessor
    handle::Ptr{Cvoid}
    schedule::Int              # CX_SCHED_* the handle was created with
    dim::Int                   # 1: scalar messages; 2 .. 64: d-dimensional linear-Gaussian messages (2, 3, 4 in registers; 5 .. 64 on the matrix cores in tiles of 16: the smallest of 16, 32, 64 that holds d, under every schedule)
    queue::Vector{CxItem}
    signals::Vector{Cortex.InferenceSignal}
end

check(h, rc) = rc == 0 ? nothing : error(unsafe_string(ccall((:cx_last_error, lib), Cstring, (Ptr{Cvoid},), h)))

# schedule: 0 flooding, 1 fused (default), 2 chain scan (paths: one cx_sweep = one update_marginals!), 3 tree (any forest: the same),
#           4 REFERENCE ORDER (ABI 3; every dim since ABI 4): ANY graph, loops included — update_marginals!(engine, ids) below becomes ONE cx_sweep_for(ids):
#           the signals Cortex.jl's own scheduler would compute for exactly this request, in its order, each from the values its rule
#           call would read (the library keeps a shadow of the readiness nibbles, driven by set_datum! / set_message! / process! / the
#           calls themselves), replayed as one graph launch; lazy like the reference (priors are re-set before a call to be fresh)
# marginals: 1 every sweep writes every marginal; 2 (chain scan, dim 2 .. 4) on demand — formed when get_marginals asks
function HipProcessor(; device = 0, dim = 1, schedule = 1, marginals = 1)
    cfg = Ref(CxConfig(sizeof(CxConfig), device, dim, schedule, marginals, 0, 0, 0))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:cx_create, lib), Int32, (Ref{CxConfig}, Ref{Ptr{Cvoid}}), cfg, out)
    rc == 0 || error(unsafe_string(ccall((:cx_last_error, lib), Cstring, (Ptr{Cvoid},), C_NULL)))
    p = HipProcessor(out[], schedule, dim, CxItem[], Cortex.InferenceSignal[])
    finalizer(q -> ccall((:cx_destroy, lib), Int32, (Ptr{Cvoid},), q.handle), p)
    return p
end

# graph ingestion through the reference's own accessors (model_engine.jl:329-391).  Directed factors (CX_FACTOR_GAUSS_LINEAR:
# x_out = a x_in + b + N(0, q), or x_out = A x_in + N(0, Q) for dim > 1) read the edge roles from Connection.label:
# role_of(engine, v, f) -> 0 (:out) | 1 (:in).  For dim > 1 params_of returns (parameter_set, 0, 0, 0) and the matrices of
# every set are supplied with set_factor_matrices! BEFORE the first sweep / batch.
function upload!(p::HipProcessor, engine::Cortex.InferenceEngine, kind_of, params_of; role_of = (engine, v, f) -> 0)
    ev, ef, er = Int64[], Int64[], Int32[]
    fids = collect(Int64, Cortex.get_factor_ids(engine))
    for f in fids, v in Cortex.get_connected_variable_ids(engine, f)
        push!(ev, v); push!(ef, f); push!(er, role_of(engine, v, f))
    end
    kinds = Int32[kind_of(Cortex.get_factor(engine, f)) for f in fids]          # CX_FACTOR_*
    params = reduce(vcat, [collect(Float64, params_of(Cortex.get_factor(engine, f))) for f in fids])  # 4 per factor
    check(p.handle, ccall((:cx_graph_create, lib), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}),
        p.handle, length(ev), ev, ef, er, length(fids), fids, kinds, params))
end

# dim > 1: (A, Q) of parameter set `set` (0-based), d x d each; the ABI is row-major, Julia column-major: pass the transposes
set_factor_matrices!(p::HipProcessor, set, A::Matrix{Float64}, Q::Matrix{Float64}) =
    check(p.handle, ccall((:cx_set_factor_matrices, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}),
                          p.handle, set, collect(transpose(A)), collect(transpose(Q))))

# where the user calls set_value!(message_to_factor(y, likelihood), datum): a Real for dim == 1, a length-d vector for dim > 1
function set_datum!(p::HipProcessor, signal::Cortex.InferenceSignal, datum)
    v = Cortex.get_variant(signal)
    check(p.handle, ccall((:cx_set_messages, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Int32, Ptr{Float64}),
                          p.handle, 1, Int64[v.variable_id], Int64[v.factor_id], 1, 1, collect(Float64, datum)))   # CX_TO_FACTOR, CX_FORM_POINT
    Cortex.set_value!(signal, datum)
end

# batched mode: collect like InferenceRequestScanner (inference_engine.jl:528-537), compute on flush.
# Every dim takes MessageToVariable / MessageToFactor / IndividualMarginal / ProductOfMessages items; JointMarginal is dim == 1.
function Cortex.process!(p::HipProcessor, engine::Cortex.InferenceEngine, variable_id, signal::Cortex.InferenceSignal)
    v = Cortex.get_variant(signal)
    item = v isa Cortex.InferenceSignalVariants.MessageToVariable ? CxItem(2, 0, v.variable_id, v.factor_id) :
           v isa Cortex.InferenceSignalVariants.MessageToFactor   ? CxItem(1, 0, v.variable_id, v.factor_id) :
           v isa Cortex.InferenceSignalVariants.IndividualMarginal ? CxItem(4, 0, v.variable_id, 0) :
           v isa Cortex.InferenceSignalVariants.ProductOfMessages  ? (issorted(v.factors_connected_to_variable) || error("ProductOfMessages: the device resolves ranges over ascending factor ids");
                                                                      CxItem(8, 0, v.variable_id, (Int64(first(v.range)) << 32) | Int64(last(v.range)))) :   # CX_ITEM_RANGE
           v isa Cortex.InferenceSignalVariants.JointMarginal      ? CxItem(16, 0, 0, v.factor_id) :
           error("The HIP processor has no rule for $(typeof(v))")
    push!(p.queue, item); push!(p.signals, signal)
    flush!(p)                       # or defer: flush once per scan wavefront
end

function flush!(p::HipProcessor)
    isempty(p.queue) && return
    check(p.handle, ccall((:cx_update_batch_async, lib), Int32, (Ptr{Cvoid}, Ptr{CxItem}, Int64), p.handle, p.queue, length(p.queue)))
    for s in p.signals
        Cortex.set_value!(s, HipValue())     # any non-UndefValue(): readiness bits evolve as in the reference (signal.jl:232-253)
    end
    empty!(p.queue); empty!(p.signals)
end
struct HipValue end                           # the payload stays in HBM; read it back with cx_get_messages / cx_get_marginals

# where the user calls set_value!(message_to_variable(x, prior), NormalMeanVariance(m, v)) (priors, seeds; dim == 1)
function set_message!(p::HipProcessor, signal::Cortex.InferenceSignal, mean, variance)
    v = Cortex.get_variant(signal)
    dir = v isa Cortex.InferenceSignalVariants.MessageToFactor ? 1 : 2                  # CX_TO_FACTOR / CX_TO_VARIABLE
    check(p.handle, ccall((:cx_set_messages, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Int32, Ptr{Float64}),
                          p.handle, 1, Int64[v.variable_id], Int64[v.factor_id], dir, 0, Float64[mean, variance]))           # CX_FORM_MOMENT
    Cortex.set_value!(signal, (mean = mean, variance = variance))
end

# damping of the fused / flooding sweeps (ABI 3): new = (1 - lambda) rule + lambda old; 0 = off
set_damping!(p::HipProcessor, lambda) = check(p.handle, ccall((:cx_set_damping, lib), Int32, (Ptr{Cvoid}, Float64), p.handle, lambda))

# whole-call mode.  Under schedule 4 the request is honoured as the reference honours it — the named variables, in the caller's order,
# only what is pending for them (src/inference_engine.jl:298-323,559-632): cx_sweep_for.  The other schedules compute every message.
function Cortex.update_marginals!(engine::Cortex.InferenceEngine{M,D,HipProcessor}, ids::Union{AbstractVector,Tuple}) where {M,D}
    p = Cortex.get_inference_request_processor(engine)
    if p.schedule == 4
        check(p.handle, ccall((:cx_sweep_for, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}), p.handle, length(ids), collect(Int64, ids)))
    else
        check(p.handle, ccall((:cx_sweep, lib), Int32, (Ptr{Cvoid}, Int32), p.handle, 1))
    end
    d = p.dim
    out = Matrix{Float64}(undef, d == 1 ? 2 : d + d * d, length(ids))           # per marginal: mean[d] then covariance[d*d]
    check(p.handle, ccall((:cx_get_marginals, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}),
                          p.handle, length(ids), collect(Int64, ids), out))
    for (i, id) in enumerate(ids)   # make the values visible as the reference's test structs (runtests.jl:31-34, :70-77)
        value = d == 1 ? (mean = out[1, i], variance = out[2, i]) : (mean = out[1:d, i], covariance = reshape(out[d+1:end, i], d, d))
        Cortex.set_value!(Cortex.get_variable_marginal(Cortex.get_variable(engine, id)), value)
    end
    return nothing
end

# ---- variational families (weak dependencies): the marginals are the state ------------------------------------------------
# A processor for the models of test/inference_engine_tests.jl:593-805 ("Mean Field", family = 2) and :807-1147
# ("Structured", family = 3).  The dependency wiring those tests' resolvers create is what the device implements, so the
# engine is built with `resolve_dependencies = false`; the host keeps only the marginal signals.
mutable struct HipVmpProcessor <: Cortex.AbstractInferenceRequestProcessor
    handle::Ptr{Cvoid}
end

function HipVmpProcessor(; device = 0, family = 3, schedule = 2)          # CX_FAMILY_VMP_STRUCTURED, CX_SCHED_CHAIN_SCAN
    cfg = Ref(CxConfig(sizeof(CxConfig), device, 1, schedule, 1, 0, family, 0))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:cx_create, lib), Int32, (Ref{CxConfig}, Ref{Ptr{Cvoid}}), cfg, out)
    rc == 0 || error(unsafe_string(ccall((:cx_last_error, lib), Cstring, (Ptr{Cvoid},), C_NULL)))
    p = HipVmpProcessor(out[])
    finalizer(q -> ccall((:cx_destroy, lib), Int32, (Ptr{Cvoid},), q.handle), p)
    return p
end

# role_of(engine, variable_id, factor_id) -> 0 (out) | 1 (mean) | 2 (precision); all factors are CX_FACTOR_NORMAL_PRECISION (3)
function upload!(p::HipVmpProcessor, engine::Cortex.InferenceEngine, role_of)
    ev, ef, er = Int64[], Int64[], Int32[]
    fids = collect(Int64, Cortex.get_factor_ids(engine))
    for f in fids, v in Cortex.get_connected_variable_ids(engine, f)
        push!(ev, v); push!(ef, f); push!(er, role_of(engine, v, f))
    end
    kinds = fill(Int32(3), length(fids))
    check(p.handle, ccall((:cx_graph_create, lib), Int32,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int32}, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}),
        p.handle, length(ev), ev, ef, er, length(fids), fids, kinds, C_NULL))
end

# where the tests call set_value!(get_variable_marginal(...), value): form 1 = datum, 3 = (mean, precision), 4 = Gamma(shape, scale)
function set_marginal!(p, variable_id, form, payload::Vector{Float64})      # HipVmpProcessor, or a HipProcessor of schedule 4 with a wiring whose messages depend on marginals
    check(p.handle, ccall((:cx_set_marginals, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int32, Ptr{Float64}),
                          p.handle, 1, Int64[variable_id], form, payload))
end

function Cortex.update_marginals!(engine::Cortex.InferenceEngine{M,D,HipVmpProcessor}, ids::Union{AbstractVector,Tuple}) where {M,D}
    p = Cortex.get_inference_request_processor(engine)
    v = collect(Int64, ids)
    check(p.handle, ccall((:cx_update_marginals, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}), p.handle, length(v), v))
    out = Matrix{Float64}(undef, 2, length(v))      # (mean, precision) or (shape, scale) per column
    check(p.handle, ccall((:cx_get_marginals, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}), p.handle, length(v), v, out))
    for (i, id) in enumerate(v)
        Cortex.set_value!(Cortex.get_variable_marginal(Cortex.get_variable(engine, id)), (out[1, i], out[2, i]))
    end
    return nothing
end

# ---- checkpoint and multi-GPU (one Julia process per GPU) ---------------------------------------------------------------
function save_state(p, path)
    n = Ref{Int64}(0)
    check(p.handle, ccall((:cx_state_bytes, lib), Int32, (Ptr{Cvoid}, Ref{Int64}), p.handle, n))
    buf = Vector{UInt8}(undef, n[])
    check(p.handle, ccall((:cx_state_export, lib), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), p.handle, buf, n[]))
    write(path, buf)
end
# factors of more than two variables (CX_FACTOR_GAUSS_LINEAR_N = 5: x_out = sum a_i x_i + b + N(0, q)): coefficients of the ROLE_IN edges
set_factor_coefficients!(p, variable_ids::Vector{Int64}, factor_ids::Vector{Int64}, a::Vector{Float64}) =
    check(p.handle, ccall((:cx_set_factor_coefficients, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}),
                          p.handle, length(a), variable_ids, factor_ids, a))

# schedule 4: a user resolver's wiring instead of the default one — one (signal, dependency, flags) triple per
# add_dependency!(signal, dependency; weak, intermediate, listen) the resolver would issue (flags: 1 weak, 2 intermediate, 4 listen = false;
# 8: signal = IndividualMarginal(v), run DefaultDependencyResolver's resolve_variable_dependencies!(v) here; 16: signal = JointMarginal(f),
# dependency = IndividualMarginal(v): link_signal_to_variable!(v, signal)).  Signals as CxItem: kind 1 MessageToFactor / 2 MessageToVariable
# with (variable_id, factor_id); 4 IndividualMarginal with variable_id; 16 JointMarginal with factor_id.  Messages that depend on marginals
# are served for CX_FACTOR_NORMAL_PRECISION (= 3) factors: the rules of test/inference_engine_tests.jl:647-689, 939-1030; the initial
# marginals and the data then go through cx_set_marginals (set_marginal! below takes any handle: pass the HipProcessor's).  Right after upload!, before any value is set.
wire!(p::HipProcessor, signals::Vector{CxItem}, dependencies::Vector{CxItem}, flags::Vector{Int32}) =
    check(p.handle, ccall((:cx_graph_wire, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{CxItem}, Ptr{CxItem}, Ptr{Int32}), p.handle, length(flags), signals, dependencies, flags))

# schedule 4 with a USER resolver: the engine was constructed with it, so its add_dependency! calls are on the signals — read them back
# (dependencies in order, weak / intermediate nibbles, the dependency's listenmask, the variables' linked signals) and hand them over.
# A variable whose signals hang off ProductOfMessages nodes was wired by the default resolver's segment tree: flag 8.  Factor side first.
# (cortex.jl_amd/wiring.py: from_engine is the same walk over the Python host mirror, exercised by tests/test_gpu_wired_vmp.py.)
function wire_from_engine!(p::HipProcessor, engine::Cortex.InferenceEngine)
    V = Cortex.InferenceSignalVariants
    sig, dep, flags = CxItem[], CxItem[], Int32[]
    name(s) = let v = Cortex.get_variant(s)
        v isa V.MessageToFactor ? CxItem(1, 0, v.variable_id, v.factor_id) :
        v isa V.MessageToVariable ? CxItem(2, 0, v.variable_id, v.factor_id) :
        v isa V.IndividualMarginal ? CxItem(4, 0, v.variable_id, 0) :
        v isa V.JointMarginal ? CxItem(16, 0, 0, v.factor_id) : error("cx_graph_wire cannot name a signal of variant $(v)")
    end
    joints, seen = Cortex.InferenceSignal[], Set{UInt}()
    note(d) = (Cortex.get_variant(d) isa V.JointMarginal && !(objectid(d) in seen)) && (push!(seen, objectid(d)); push!(joints, d))
    function export_signal(s)
        for (i, d) in enumerate(Cortex.get_dependencies(s))
            note(d)
            fl = Int32(0)
            Cortex.is_dependency_weak(s.dependencies_props, i) && (fl |= Int32(1))
            Cortex.is_dependency_intermediate(s.dependencies_props, i) && (fl |= Int32(2))
            d.listenmask[findfirst(l -> l === s, Cortex.get_listeners(d))] || (fl |= Int32(4))
            push!(sig, name(s)); push!(dep, name(d)); push!(flags, fl)
        end
    end
    for f in Cortex.get_factor_ids(engine), v in Cortex.get_connected_variable_ids(engine, f)
        export_signal(Cortex.get_connection_message_to_variable(engine, v, f))
    end
    for v in Cortex.get_variable_ids(engine), ls in Cortex.get_variable_linked_signals(Cortex.get_variable(engine, v))
        note(ls)
    end
    done = 0
    while done < length(joints); done += 1; export_signal(joints[done]); end
    for v in Cortex.get_variable_ids(engine)
        variable = Cortex.get_variable(engine, v)
        marg = Cortex.get_variable_marginal(variable)
        own = vcat([marg], [Cortex.get_connection_message_to_factor(engine, v, f) for f in Cortex.get_connected_factor_ids(engine, v)])
        if any(Cortex.get_variant(d) isa V.ProductOfMessages for s in own for d in Cortex.get_dependencies(s))
            push!(sig, name(marg)); push!(dep, name(marg)); push!(flags, Int32(8))
        else
            foreach(export_signal, own)
        end
        for ls in Cortex.get_variable_linked_signals(variable)
            push!(sig, name(ls)); push!(dep, name(marg)); push!(flags, Int32(16))
        end
    end
    while done < length(joints); done += 1; export_signal(joints[done]); end
    wire!(p, sig, dep, flags)
end

# schedule 4: the executions of the last call in the reference's order — what a `trace = true` engine records as
# TracedInferenceExecution.signal (src/inference_engine.jl:650-862) — as (kind, variable_id, factor_id | range) items
function reference_trace(p)
    n = Ref{Int64}(0)
    check(p.handle, ccall((:cx_ref_trace, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{CxItem}, Ref{Int64}), p.handle, 0, C_NULL, n))
    items = Vector{CxItem}(undef, n[])
    check(p.handle, ccall((:cx_ref_trace, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{CxItem}, Ref{Int64}), p.handle, n[], items, n))
    return items
end
function reference_plan_stats(p)       # stages, launches, executions, messages, passes, plans kept, calls that replayed a plan, calls that ran the scheduler
    out = zeros(Int64, 8)
    check(p.handle, ccall((:cx_ref_plan_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    return out
end

function tree_plan_stats(p)            # CX_SCHED_TREE (schedule = 3): depth, stages, items, k-ary entries, components, up, down, marginals
    out = zeros(Int64, 8)
    check(p.handle, ccall((:cx_tree_plan_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    return out
end

function message_health(p)             # numerical guards as counters: defined, undefined, negative precision, non-finite (messages into non-observed variables)
    out = zeros(Int64, 4)
    check(p.handle, ccall((:cx_message_health, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    return out
end

function tree_heavy_path_stats(p)      # light depths, paths, variables on no path, launches per sweep (zeros: the level schedule is in use)
    out = zeros(Int64, 4)
    check(p.handle, ccall((:cx_tree_heavy_path_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    return out
end

# the plan of the dim 64 chain-scan schedule: (links per block, fan, levels, potentials, compositions, rules, launches, device bytes)
function chain_plan_stats(p)
    out = zeros(Int64, 8)
    check(p.handle, ccall((:cx_chain_plan_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    out
end

# the XCD-resident cluster of reference-order plans: (1 ready / 0 not prepared / -1 off, workgroups per launch, calls finished on plain
# launches after a barrier of the cluster timed out — such a call still returns its results —, 1 when the last call ran on it)
function cluster_stats(p)
    out = zeros(Int64, 4)
    check(p.handle, ccall((:cx_cluster_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    out
end

# the scalar chain scan as one launch: (1 ready / 0 not prepared / -1 off, launches of that form, 0, 0)
function chain_scan_stats(p)
    out = zeros(Int64, 4)
    check(p.handle, ccall((:cx_chain_scan_stats, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), p.handle, out))
    out
end

load_state!(p, path) = (buf = read(path); check(p.handle, ccall((:cx_state_import, lib), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int64), p.handle, buf, length(buf))))

# deep halo: `exchange_every` plain sweeps between two state exchanges issued by the library over RCCL
function sweep_partitioned!(p, n_sweeps, exchange_every)
    for k in 0:(n_sweeps - 1)
        k % exchange_every == 0 && check(p.handle, ccall((:cx_halo_state_exchange, lib), Int32, (Ptr{Cvoid},), p.handle))
        check(p.handle, ccall((:cx_sweep, lib), Int32, (Ptr{Cvoid}, Int32), p.handle, 1))
    end
end
end
