"""cortex.jl_amd — the MI355X-native sum-product sweep behind Cortex.jl's InferenceEngine/processor API.

Import as ``import cortex.jl_amd as cx`` (the top-level ``cortex`` package is a loader shim for
this directory, whose name contains a dot).  The device path has no CPU fallback.
"""
from . import _lib
from ._lib import CortexHipError
from .device import DeviceGraph
from . import synth
from . import wiring
from .signal import (Signal, UndefValue, UndefVariant, add_dependency, compute, get_dependencies, get_listeners,
                     get_value, get_variant, is_computed, is_pending, isa_variant, process_dependencies, set_value,
                     set_variant)
from .inference_signal import InferenceSignal, InferenceSignalVariants, create_inference_signal
from .model_engine import (BipartiteFactorGraph, Connection, Factor, UnsupportedModelEngineError, Variable,
                           get_connection_message_to_factor, get_connection_message_to_variable,
                           get_factor_functional_form, get_variable_linked_signals, get_variable_marginal,
                           is_engine_supported, link_signal_to_variable)
from .dependencies import (AbstractDependencyResolver, DefaultDependencyResolver, form_segment_tree_dependency,
                           resolve_dependencies)
from .inference_engine import (AbstractInferenceRequestProcessor, InferenceEngine, InferenceRequestScanner,
                               request_inference_for, scan_inference_request, update_marginals)
from .hip_processor import (Beta, Gamma, GaussianAdditive, GaussianLinear, HipProcessor, HipValue, HipVmpProcessor, HipVmpValue,
                            MvGaussianLinear, MvNormalMeanCovariance, MvNormalMeanPrecision, NormalMeanPrecision, NormalMeanVariance,
                            NormalPrecisionFactor, run_wavefronts)

__all__ = [n for n in dir() if not n.startswith("__")]
