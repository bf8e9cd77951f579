"""Mirror of src/inference_signal.jl: the five variant tags and the InferenceSignal constructor."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from .signal import Signal, UndefValue


class InferenceSignalVariants:
    @dataclass(frozen=True)
    class Unspecified:                 # inference_signal.jl:16
        pass

    @dataclass(frozen=True)
    class MessageToFactor:             # :29-32
        variable_id: int
        factor_id: int

    @dataclass(frozen=True)
    class MessageToVariable:           # :45-48
        variable_id: int
        factor_id: int

    @dataclass(frozen=True)
    class ProductOfMessages:           # :62-66 (range is 1-based inclusive, as in Julia)
        variable_id: int
        range: Tuple[int, int]
        factors_connected_to_variable: Tuple[int, ...]

    @dataclass(frozen=True)
    class IndividualMarginal:          # :78-80
        variable_id: int

    @dataclass(frozen=True)
    class JointMarginal:               # :93-96
        factor_id: int
        variable_ids: Tuple[int, ...]


InferenceSignal = Signal               # inference_signal.jl:129 (Signal{Any, InferenceSignalVariant})


def create_inference_signal() -> Signal:
    """inference_signal.jl:140"""
    return Signal(UndefValue(), InferenceSignalVariants.Unspecified())
