"""cortex.jl_amd — the MI355X-native sum-product sweep behind Cortex.jl's InferenceEngine/processor API.

Import as ``import cortex.jl_amd as cx`` (the top-level ``cortex`` package is a loader shim for
this directory, whose name contains a dot).  The device path has no CPU fallback.
"""
from . import _lib
from ._lib import CortexHipError
from .device import DeviceGraph
from . import synth

__all__ = ["_lib", "CortexHipError", "DeviceGraph", "synth"]
