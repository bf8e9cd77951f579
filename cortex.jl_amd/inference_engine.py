"""Mirror of src/inference_engine.jl:1-632 — the scheduler the host keeps, and the processor plugin API that the
HIP sweep drops in behind.  `update_marginals!` → update_marginals, `process!` → process (a processor method)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

from .dependencies import AbstractDependencyResolver, DefaultDependencyResolver, resolve_dependencies
from .inference_signal import InferenceSignalVariants as V
from .model_engine import (get_connection_message_to_factor, get_connection_message_to_variable,
                           get_variable_linked_signals, get_variable_marginal, throw_if_engine_unsupported)
from .signal import Signal, compute, get_value, is_pending, process_dependencies, set_variant


@dataclass
class InferenceEngineWarning:          # inference_engine.jl:11-14
    description: str
    context: Any


class AbstractInferenceRequestProcessor:
    """inference_engine.jl:331-509.  Subclasses implement the five rules; the defaults raise, as the reference's
    `error("The function ... is not implemented for the processor of type ...")`."""

    def _missing(self, name):
        raise NotImplementedError(f"The function `{name}` is not implemented for the processor of type {type(self).__name__}")

    def compute_message_to_variable(self, engine, variant, signal, dependencies):
        self._missing("compute_message_to_variable!")

    def compute_message_to_factor(self, engine, variant, signal, dependencies):
        self._missing("compute_message_to_factor!")

    def compute_individual_marginal(self, engine, variant, signal, dependencies):
        self._missing("compute_individual_marginal!")

    def compute_product_of_messages(self, engine, variant, signal, dependencies):
        self._missing("compute_product_of_messages!")

    def compute_joint_marginal(self, engine, variant, signal, dependencies):
        self._missing("compute_joint_marginal!")

    def process(self, engine, variable_id, dependency: Signal):
        """process!, inference_engine.jl:479-509: variant `isa` chain → rule → set_value! (through compute!)."""
        def strategy(signal, dependencies):
            variant = signal.variant
            if isinstance(variant, V.MessageToVariable):
                return self.compute_message_to_variable(engine, variant, signal, dependencies)
            if isinstance(variant, V.MessageToFactor):
                return self.compute_message_to_factor(engine, variant, signal, dependencies)
            if isinstance(variant, V.IndividualMarginal):
                return self.compute_individual_marginal(engine, variant, signal, dependencies)
            if isinstance(variant, V.ProductOfMessages):
                return self.compute_product_of_messages(engine, variant, signal, dependencies)
            if isinstance(variant, V.JointMarginal):
                return self.compute_joint_marginal(engine, variant, signal, dependencies)
            raise RuntimeError(f"Unprocessed signal variant: {signal.variant!r}")
        compute(strategy, dependency)

    # hooks a batching processor overrides (no-ops for per-signal processors)
    def flush(self, engine):
        pass


class InferenceRequestScanner(AbstractInferenceRequestProcessor):
    """inference_engine.jl:528-537: collects instead of computing."""

    def __init__(self):
        self.signals: List[Signal] = []

    def process(self, engine, variable_id, dependency):
        self.signals.append(dependency)


@dataclass
class TracedInferenceExecution:        # inference_engine.jl:650-700 (timing fields omitted: not on the path)
    variable_id: Any
    signal: Signal
    value_before_execution: Any
    value_after_execution: Any


@dataclass
class TracedInferenceRound:
    executions: List[TracedInferenceExecution] = field(default_factory=list)


@dataclass
class TracedInferenceRequest:
    variable_ids: Any
    rounds: List[TracedInferenceRound] = field(default_factory=list)


@dataclass
class InferenceEngineTracer:
    inference_requests: List[TracedInferenceRequest] = field(default_factory=list)


@dataclass
class InferenceRequest:                # inference_engine.jl:265-270
    engine: "InferenceEngine"
    variable_ids: Sequence
    marginals: List[Signal]
    readines_status: List[bool]


class InferenceEngine:
    """inference_engine.jl:53-89"""

    def __init__(self, *, model_engine, dependency_resolver: Optional[AbstractDependencyResolver] = None,
                 inference_request_processor: Optional[AbstractInferenceRequestProcessor] = None,
                 prepare_signals_metadata: bool = True, resolve_dependencies: bool = True, trace: bool = False):
        self.model_engine = throw_if_engine_unsupported(model_engine)
        self.dependency_resolver = dependency_resolver if dependency_resolver is not None else DefaultDependencyResolver()
        if not isinstance(self.dependency_resolver, AbstractDependencyResolver):
            raise TypeError("dependency_resolver must be an AbstractDependencyResolver")          # convert(...) at :69
        self.inference_request_processor = (inference_request_processor if inference_request_processor is not None
                                            else InferenceRequestScanner())
        if not isinstance(self.inference_request_processor, AbstractInferenceRequestProcessor):
            raise TypeError("inference_request_processor must be an AbstractInferenceRequestProcessor")  # :70
        self.tracer = InferenceEngineTracer() if trace else None
        self.warnings: List[InferenceEngineWarning] = []
        if prepare_signals_metadata:
            set_signals_variants(self)
        if resolve_dependencies:
            globals()["resolve_dependencies"](self.dependency_resolver, self)
        attach = getattr(self.inference_request_processor, "attach", None)
        if attach is not None:         # build hook: a device-backed processor flattens the graph here
            attach(self)

    def __repr__(self):
        return f"InferenceEngine(trace = {'true' if self.tracer is not None else 'false'})"

    # accessor aliases, inference_engine.jl:119-205
    def get_model_engine(self):
        return self.model_engine

    def get_inference_request_processor(self):
        return self.inference_request_processor

    def get_trace(self):
        return self.tracer

    def get_warnings(self):
        return self.warnings

    def add_warning(self, description, context):
        self.warnings.append(InferenceEngineWarning(description, context))

    def get_variable(self, variable_id):
        return self.model_engine.get_variable(variable_id)

    def get_variable_ids(self):
        return self.model_engine.get_variable_ids()

    def get_factor(self, factor_id):
        return self.model_engine.get_factor(factor_id)

    def get_factor_ids(self):
        return self.model_engine.get_factor_ids()

    def get_connection(self, variable_id, factor_id):
        return self.model_engine.get_connection(variable_id, factor_id)

    def get_connection_message_to_variable(self, variable_id, factor_id):
        return get_connection_message_to_variable(self.get_connection(variable_id, factor_id))

    def get_connection_message_to_factor(self, variable_id, factor_id):
        return get_connection_message_to_factor(self.get_connection(variable_id, factor_id))

    def get_connected_variable_ids(self, factor_id):
        return self.model_engine.get_connected_variable_ids(factor_id)

    def get_connected_factor_ids(self, variable_id):
        return self.model_engine.get_connected_factor_ids(variable_id)


def set_signals_variants(engine: InferenceEngine):
    """inference_engine.jl:228-247"""
    for variable_id in engine.get_variable_ids():
        set_variant(get_variable_marginal(engine.get_variable(variable_id)), V.IndividualMarginal(variable_id))
    for factor_id in engine.get_factor_ids():
        for variable_id in engine.get_connected_variable_ids(factor_id):
            c = engine.get_connection(variable_id, factor_id)
            set_variant(get_connection_message_to_factor(c), V.MessageToFactor(variable_id, factor_id))
            set_variant(get_connection_message_to_variable(c), V.MessageToVariable(variable_id, factor_id))


def _as_ids(variable_id_or_ids):
    if isinstance(variable_id_or_ids, (list, tuple)):
        return variable_id_or_ids
    try:
        import numpy as np
        if isinstance(variable_id_or_ids, np.ndarray):
            return [int(x) for x in variable_id_or_ids]
    except ImportError:  # pragma: no cover
        pass
    return (variable_id_or_ids,)


def request_inference_for(engine: InferenceEngine, variable_id_or_ids) -> InferenceRequest:
    """inference_engine.jl:298-323"""
    ids = _as_ids(variable_id_or_ids)
    marginals = []
    for variable_id in ids:
        variable = engine.get_variable(variable_id)
        marginal = get_variable_marginal(variable)
        for dependency in marginal.dependencies:
            dependency.is_potentially_pending, dependency._is_pending = True, False
        for linked in get_variable_linked_signals(variable):
            linked.is_potentially_pending, linked._is_pending = True, False
        marginals.append(marginal)
    return InferenceRequest(engine, ids, marginals, [False] * len(ids))


def process_inference_request(processor, request: InferenceRequest, variable_id, marginal, trace=None) -> bool:
    """inference_engine.jl:512-525"""
    def f(dependency):
        if is_pending(dependency):
            _traced(trace, variable_id, dependency, lambda: processor.process(request.engine, variable_id, dependency))
            return True
        return False
    return process_dependencies(f, marginal, retry=True)


def scan_inference_request(request: InferenceRequest) -> List[Signal]:
    """inference_engine.jl:540-546"""
    scanner = InferenceRequestScanner()
    for variable_id, marginal in zip(request.variable_ids, request.marginals):
        process_inference_request(scanner, request, variable_id, marginal)
    return scanner.signals


def _traced(round_trace, variable_id, signal, fn):
    """trace_inference_execution, inference_engine.jl:830-862"""
    if round_trace is None:
        return fn()
    before = get_value(signal)
    fn()
    round_trace.executions.append(TracedInferenceExecution(variable_id, signal, before, get_value(signal)))


def update_marginals(engine: InferenceEngine, variable_id_or_ids):
    """update_marginals!, inference_engine.jl:559-632.  A processor may take the whole call over (the analogue of a
    Julia method specialised on the processor type): it then defines `update_marginals(engine, ids)`."""
    ids = _as_ids(variable_id_or_ids)
    processor = engine.get_inference_request_processor()
    takeover = getattr(processor, "update_marginals", None)
    if takeover is not None:
        if engine.tracer is None:
            if takeover(engine, ids):
                return None
        else:
            # a whole-call takeover still leaves a trace: one request, one round, the requested marginals in request order
            # with the values they held before and after (the reference's per-signal records, inference_engine.jl:650-700,
            # have no counterpart when a launch computes every message at once)
            marginals = [get_variable_marginal(engine.get_variable(v)) for v in ids]
            before = [get_value(m) for m in marginals]
            if takeover(engine, ids):
                req = TracedInferenceRequest(ids)
                rnd = TracedInferenceRound([TracedInferenceExecution(v, m, b, get_value(m)) for v, m, b in zip(ids, marginals, before)])
                req.rounds.append(rnd)
                engine.tracer.inference_requests.append(req)
                return None
    request = request_inference_for(engine, ids)
    req_trace = None
    if engine.tracer is not None:
        req_trace = TracedInferenceRequest(ids)

    def new_round():
        return TracedInferenceRound() if req_trace is not None else None

    def end_round(r):
        if r is not None and r.executions:     # empty rounds are dropped (:818)
            req_trace.rounds.append(r)

    n = len(ids)
    should_continue, is_reverse = True, False
    while should_continue:
        cont = False
        rt = new_round()
        order = range(n - 1, -1, -1) if is_reverse else range(n)
        for i in order:
            if not request.readines_status[i]:
                marginal = request.marginals[i]
                did = process_inference_request(processor, request, ids[i], marginal, trace=rt)
                processor.flush(engine)
                if is_pending(marginal):
                    request.readines_status[i] = True
                cont = cont or did
        end_round(rt)
        is_reverse = not is_reverse
        should_continue = cont
    rt = new_round()
    for variable_id, marginal in zip(request.variable_ids, request.marginals):
        if is_pending(marginal):
            _traced(rt, variable_id, marginal, lambda: processor.process(engine, variable_id, marginal))
        for linked in get_variable_linked_signals(engine.get_variable(variable_id)):
            if not is_pending(linked):
                continue
            _traced(rt, variable_id, linked, lambda linked=linked: processor.process(engine, variable_id, linked))
    processor.flush(engine)
    end_round(rt)
    if engine.tracer is not None:
        engine.tracer.inference_requests.append(req_trace)
    return None
