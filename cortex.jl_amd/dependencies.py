"""Mirror of src/dependencies.jl: the default belief-propagation dependency resolver (incl. the segment tree)."""
from __future__ import annotations

from .inference_signal import InferenceSignalVariants, create_inference_signal
from .model_engine import get_variable_marginal
from .signal import add_dependency, get_listeners, set_variant


class AbstractDependencyResolver:      # dependencies.jl:1
    def resolve_factor_dependencies(self, engine, factor_id):
        raise NotImplementedError

    def resolve_variable_dependencies(self, engine, variable_id):
        raise NotImplementedError


def resolve_dependencies(resolver: AbstractDependencyResolver, engine):
    """dependencies.jl:5-15 — factors first, then variables."""
    for factor_id in engine.get_factor_ids():
        resolver.resolve_factor_dependencies(engine, factor_id)
    for variable_id in engine.get_variable_ids():
        resolver.resolve_variable_dependencies(engine, variable_id)


class DefaultDependencyResolver(AbstractDependencyResolver):  # dependencies.jl:3
    def resolve_factor_dependencies(self, engine, factor_id):
        """dependencies.jl:17-31"""
        ids = engine.get_connected_variable_ids(factor_id)
        for v1 in ids:
            for v2 in ids:
                if v1 != v2:
                    add_dependency(engine.get_connection_message_to_variable(v1, factor_id),
                                   engine.get_connection_message_to_factor(v2, factor_id))

    def resolve_variable_dependencies(self, engine, variable_id):
        """dependencies.jl:33-126"""
        factors = engine.get_connected_factor_ids(variable_id)
        marginal = get_variable_marginal(engine.get_variable(variable_id))
        n = len(factors)
        if n == 0:
            engine.add_warning("Variable has no connected factors", variable_id)
            return
        if n < 2:
            add_dependency(marginal, engine.get_connection_message_to_variable(variable_id, factors[0]), intermediate=True)
            return
        if n <= 5:
            for f in factors:
                add_dependency(marginal, engine.get_connection_message_to_variable(variable_id, f), intermediate=True)
                m2f = engine.get_connection_message_to_factor(variable_id, f)
                if get_listeners(m2f):
                    for other in factors:
                        if other != f:
                            add_dependency(m2f, engine.get_connection_message_to_variable(variable_id, other), intermediate=True)
            return
        mid = n // 2
        left = form_segment_tree_dependency(engine, (0, mid), factors, variable_id)
        right = form_segment_tree_dependency(engine, (mid, n), factors, variable_id)
        _cross(engine, variable_id, factors, (0, mid), right)
        _cross(engine, variable_id, factors, (mid, n), left)
        add_dependency(marginal, left, intermediate=True)
        add_dependency(marginal, right, intermediate=True)


def _cross(engine, variable_id, factors, rng, other_side):
    for f in factors[rng[0]:rng[1]]:
        m2f = engine.get_connection_message_to_factor(variable_id, f)
        if get_listeners(m2f):
            add_dependency(m2f, other_side, intermediate=True)


def form_segment_tree_dependency(engine, rng, factors, variable_id):
    """dependencies.jl:128-173; rng is a 0-based half-open (lo, hi) over `factors`."""
    lo, hi = rng
    assert hi - lo >= 1
    if hi - lo == 1:
        return engine.get_connection_message_to_variable(variable_id, factors[lo])
    mid = lo + (hi - lo) // 2
    left = form_segment_tree_dependency(engine, (lo, mid), factors, variable_id)
    right = form_segment_tree_dependency(engine, (mid, hi), factors, variable_id)
    _cross(engine, variable_id, factors, (lo, mid), right)
    _cross(engine, variable_id, factors, (mid, hi), left)
    inter = create_inference_signal()
    set_variant(inter, InferenceSignalVariants.ProductOfMessages(variable_id, (lo + 1, hi), tuple(factors)))
    add_dependency(inter, left, intermediate=True)
    add_dependency(inter, right, intermediate=True)
    return inter
