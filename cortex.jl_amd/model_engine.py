"""Mirror of src/model_engine.jl (Variable / Factor / Connection + the 7 graph accessors) and a minimal stand-in for
the un-vendored BipartiteFactorGraphs.jl backend that ext/BipartiteFactorGraphsExt adapts: one shared 1-based id
sequence for variables and factors, neighbours iterated in ascending id."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List

from .inference_signal import create_inference_signal
from .signal import Signal


@dataclass
class Variable:                        # model_engine.jl:30-35
    name: str
    index: Any = None
    marginal: Signal = field(default_factory=create_inference_signal)
    linked_signals: List[Signal] = field(default_factory=list)


@dataclass
class Factor:                          # model_engine.jl:119-122
    functional_form: Any
    local_marginals: List[Signal] = field(default_factory=list)


@dataclass
class Connection:                      # model_engine.jl:181-186
    label: str
    index: int = 0
    message_to_variable: Signal = field(default_factory=create_inference_signal)
    message_to_factor: Signal = field(default_factory=create_inference_signal)


def get_variable_marginal(v: Variable) -> Signal:
    return v.marginal


def get_variable_linked_signals(v: Variable):
    return v.linked_signals


def link_signal_to_variable(v: Variable, s: Signal):
    v.linked_signals.append(s)


def get_factor_functional_form(f: Factor):
    return f.functional_form


def get_factor_local_marginals(f: Factor):          # model_engine.jl:141-143
    return f.local_marginals


def add_local_marginal_to_factor(f: Factor, local_marginal: Signal):   # model_engine.jl:150-153
    f.local_marginals.append(local_marginal)


def get_connection_message_to_variable(c: Connection) -> Signal:
    return c.message_to_variable


def get_connection_message_to_factor(c: Connection) -> Signal:
    return c.message_to_factor


class UnsupportedModelEngineError(Exception):
    """model_engine.jl:252-266"""

    def __init__(self, model_engine, missing_function=None):
        self.model_engine, self.missing_function = model_engine, missing_function
        if missing_function is None:
            msg = f"The model engine of type `{type(model_engine).__name__}` is not supported."
        else:
            msg = (f"The model engine of type `{type(model_engine).__name__}` does not implement the function "
                   f"`{missing_function}`.")
        super().__init__(msg)


REQUIRED_ACCESSORS = ("get_variable", "get_factor", "get_variable_ids", "get_factor_ids", "get_connection",
                      "get_connected_variable_ids", "get_connected_factor_ids")   # model_engine.jl:329-391


def is_engine_supported(engine) -> bool:
    """model_engine.jl:310: the trait; here: the 7 accessors exist."""
    return all(callable(getattr(engine, n, None)) for n in REQUIRED_ACCESSORS)


def throw_if_engine_unsupported(engine):
    """model_engine.jl:319-321"""
    if not is_engine_supported(engine):
        missing = [n for n in REQUIRED_ACCESSORS if not callable(getattr(engine, n, None))]
        raise UnsupportedModelEngineError(engine, missing[0] if len(missing) < len(REQUIRED_ACCESSORS) else None)
    return engine


class BipartiteFactorGraph:
    """What Cortex's tests build with BipartiteFactorGraph(Variable, Factor, Connection) (e.g.
    test/inference_engine_tests.jl:436-453): add_variable!/add_factor! return ids from one counter."""

    def __init__(self):
        self._n = 0
        self._var: Dict[int, Variable] = {}
        self._fac: Dict[int, Factor] = {}
        self._edge: Dict[tuple, Connection] = {}
        self._nbr: Dict[int, List[int]] = {}

    def add_variable(self, v: Variable) -> int:
        self._n += 1
        self._var[self._n] = v
        self._nbr[self._n] = []
        return self._n

    def add_factor(self, f: Factor) -> int:
        self._n += 1
        self._fac[self._n] = f
        self._nbr[self._n] = []
        return self._n

    def add_edge(self, variable_id: int, factor_id: int, c: Connection):
        if variable_id not in self._var or factor_id not in self._fac:
            raise KeyError(f"add_edge!: ({variable_id}, {factor_id}) is not a (variable, factor) pair")
        self._edge[(variable_id, factor_id)] = c
        for a, b in ((variable_id, factor_id), (factor_id, variable_id)):
            lst = self._nbr[a]
            lst.append(b)
            lst.sort()

    # the 7 accessors (ext/BipartiteFactorGraphsExt/BipartiteFactorGraphsExt.jl:22-48)
    def get_variable(self, variable_id: int) -> Variable:
        return self._var[variable_id]

    def get_factor(self, factor_id: int) -> Factor:
        return self._fac[factor_id]

    def get_variable_ids(self):
        return sorted(self._var)

    def get_factor_ids(self):
        return sorted(self._fac)

    def get_connection(self, variable_id: int, factor_id: int) -> Connection:
        return self._edge[(variable_id, factor_id)]

    def get_connected_variable_ids(self, factor_id: int):
        return list(self._nbr[factor_id])

    def get_connected_factor_ids(self, variable_id: int):
        return list(self._nbr[variable_id])
