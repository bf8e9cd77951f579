"""ctypes binding of include/cortex_hip.h — the same entry points a Julia `ccall` shim binds
(INTEGRATION.md).  There is no CPU fallback: if the HIP library is missing the import of the
device path fails loudly."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CORTEX_HIP_LIB", os.path.join(HERE, "libcortex_hip.so"))  # override: A/B builds of the same ABI

# mirrors of the #defines in include/cortex_hip.h
ABI_VERSION = 4
OK = 0
ERR_INVALID_ARGUMENT, ERR_NOT_FOUND, ERR_UNSUPPORTED, ERR_STATE, ERR_DEVICE, ERR_NO_DEVICE, ERR_OUT_OF_MEMORY = (
    -1, -2, -3, -4, -5, -6, -7)
TO_FACTOR, TO_VARIABLE = 1, 2
ITEM_MESSAGE_TO_FACTOR, ITEM_MESSAGE_TO_VARIABLE, ITEM_INDIVIDUAL_MARGINAL = 1, 2, 4
ITEM_PRODUCT_OF_MESSAGES, ITEM_JOINT_MARGINAL = 8, 16
FORM_MOMENT, FORM_POINT, FORM_NATURAL, FORM_MEAN_PRECISION, FORM_GAMMA = 0, 1, 2, 3, 4
FACTOR_OPAQUE, FACTOR_GAUSS_ADDITIVE, FACTOR_GAUSS_LINEAR, FACTOR_NORMAL_PRECISION, FACTOR_BERNOULLI, FACTOR_GAUSS_LINEAR_N = 0, 1, 2, 3, 4, 5
NPARAM = 4
ROLE_OUT, ROLE_IN, ROLE_PRECISION = 0, 1, 2
SCHED_FLOODING, SCHED_FUSED, SCHED_CHAIN_SCAN, SCHED_TREE, SCHED_REFERENCE = 0, 1, 2, 3, 4
FAMILY_GAUSSIAN, FAMILY_NATURAL2, FAMILY_VMP_MEAN_FIELD, FAMILY_VMP_STRUCTURED = 0, 1, 2, 3
VMP_ALL_NORMAL, VMP_ALL_PRECISION = -1, -2
WIRE_WEAK, WIRE_INTERMEDIATE, WIRE_NO_LISTEN, WIRE_DEFAULT_VARIABLE, WIRE_LINK = 1, 2, 4, 8, 16
KERNEL_VAR_TO_FACTOR, KERNEL_FACTOR_TO_VAR, KERNEL_FUSED, KERNEL_BATCH, KERNEL_BIG_VAR = 0, 1, 2, 3, 4
KERNEL_HALO_BEGIN, KERNEL_HALO_END = 5, 6
KERNEL_COUNT = 8


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32), ("schedule", C.c_int32),
                ("compute_marginals_in_sweep", C.c_int32), ("materialize_messages_to_factor", C.c_int32),
                ("family", C.c_int32), ("reserved", C.c_int32)]


class Item(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("variable_id", C.c_int64), ("factor_id", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("n_variables", C.c_int64), ("n_factors", C.c_int64), ("n_edges", C.c_int64),
                ("n_messages_per_sweep", C.c_int64), ("n_slices", C.c_int64), ("n_big_variables", C.c_int64),
                ("n_slots", C.c_int64), ("device_bytes", C.c_int64), ("sweeps_done", C.c_int64)]


_i32, _i64, _dbl, _vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p
_pi32, _pi64, _pd = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)

# name -> (restype, argtypes): every symbol include/cortex_hip.h declares
SIGNATURES = {
    "cx_version": (_i32, []),
    "cx_create": (_i32, [C.POINTER(Config), C.POINTER(_vp)]),
    "cx_destroy": (_i32, [_vp]),
    "cx_last_error": (C.c_char_p, [_vp]),
    "cx_sync": (_i32, [_vp]),
    "cx_set_stream": (_i32, [_vp, _vp]),
    "cx_graph_create": (_i32, [_vp, _i64, _pi64, _pi64, _pi32, _i64, _pi64, _pi32, _pd]),
    "cx_set_factor_matrices": (_i32, [_vp, _i64, _pd, _pd]),
    "cx_graph_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "cx_set_factor_coefficients": (_i32, [_vp, _i64, _pi64, _pi64, _pd]),
    "cx_set_factor_edge_sets": (_i32, [_vp, _i64, _pi64, _pi64, _pi64]),
    "cx_edge_index": (_i32, [_vp, _i64, _pi64, _pi64, _pi64]),
    "cx_payload_doubles": (_i64, [_i32, _i32]),
    "cx_set_messages": (_i32, [_vp, _i64, _pi64, _pi64, _i32, _i32, _pd]),
    "cx_get_messages": (_i32, [_vp, _i64, _pi64, _pi64, _i32, _i32, _pd]),
    "cx_seed_messages": (_i32, [_vp, _i32, _dbl, _dbl]),
    "cx_get_marginals": (_i32, [_vp, _i64, _pi64, _pd]),
    "cx_update_batch": (_i32, [_vp, C.POINTER(Item), _i64]),
    "cx_update_batch_async": (_i32, [_vp, C.POINTER(Item), _i64]),
    "cx_get_products": (_i32, [_vp, _i64, _pi64, _pi32, _pi32, _i32, _pd]),
    "cx_get_joint_marginals": (_i32, [_vp, _i64, _pi64, _pd]),
    "cx_sweep": (_i32, [_vp, _i32]),
    "cx_sweep_for": (_i32, [_vp, _i64, _pi64]),
    "cx_set_damping": (_i32, [_vp, _dbl]),
    "cx_cluster_stats": (_i32, [_vp, _pi64]),
    "cx_graph_wire": (_i32, [_vp, _i64, C.POINTER(Item), C.POINTER(Item), _pi32]),
    "cx_ref_plan_stats": (_i32, [_vp, _pi64]),
    "cx_ref_trace": (_i32, [_vp, _i64, C.POINTER(Item), _pi64]),
    "cx_residual": (_i32, [_vp, _pd]),
    "cx_message_health": (_i32, [_vp, _pi64]),
    "cx_halo_configure": (_i32, [_vp, _i64, _pi64, _pi64, _i64, _pi64, _pi64]),
    "cx_halo_buffers": (_i32, [_vp, C.POINTER(_vp), _pi64, C.POINTER(_vp), _pi64]),
    "cx_halo_set_buffers": (_i32, [_vp, _vp, _vp]),
    "cx_sweep_begin": (_i32, [_vp]),
    "cx_sweep_main": (_i32, [_vp]),
    "cx_sweep_end": (_i32, [_vp]),
    "cx_comm_unique_id": (_i32, [_vp]),
    "cx_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "cx_halo_peers": (_i32, [_vp, _i32, _pi32, _pi64, _pi64, _pi64, _pi64]),
    "cx_sweep_exchange": (_i32, [_vp, _i32]),
    "cx_sweep_until": (_i32, [_vp, _dbl, _i32, _i32, _pi32, _pd]),
    "cx_halo_configure_state": (_i32, [_vp, _i64, _pi64, _pi64, _i64, _pi64, _pi64]),
    "cx_halo_set_layers": (_i32, [_vp, _i64, _pi64, _pi32, _i32]),
    "cx_halo_state_pack": (_i32, [_vp]),
    "cx_halo_state_unpack": (_i32, [_vp]),
    "cx_halo_state_exchange": (_i32, [_vp]),
    "cx_halo_exchange_sweep": (_i32, [_vp, _i32]),
    "cx_halo_ipc_alloc": (_i32, [_vp, C.c_char_p, C.POINTER(C.c_void_p), _pi64]),
    "cx_halo_ipc_connect": (_i32, [_vp, _i32, C.c_char_p, C.c_void_p, _i32, _i64, _i64]),
    "cx_halo_ipc_exchange": (_i32, [_vp]),
    "cx_halo_ipc_exchange_sweep": (_i32, [_vp, _i32]),
    "cx_halo_ipc_batch": (_i32, [_vp, _i32]),
    "cx_halo_ipc_set_fused": (_i32, [_vp, _i32]),
    "cx_halo_ipc_push": (_i32, [_vp]),
    "cx_halo_ipc_unpack": (_i32, [_vp]),
    "cx_halo_ipc_status": (_i32, [_vp, _pi32, _pi64]),
    "cx_halo_ipc_set_timeout": (_i32, [_vp, C.c_double]),
    "cx_chain_block_maps": (_i32, [_vp, _pd, _pd, _pd, _pd, _pi64, _pi64, _pi64]),
    "cx_chain_plan_stats": (_i32, [_vp, _pi64]),
    "cx_chain_scan_stats": (_i32, [_vp, _pi64]),
    "cx_tree_plan_stats": (_i32, [_vp, _pi64]),
    "cx_tree_heavy_path_stats": (_i32, [_vp, _pi64]),
    "cx_set_marginals": (_i32, [_vp, _i64, _pi64, _i32, _pd]),
    "cx_update_marginals": (_i32, [_vp, _i64, _pi64]),
    "cx_state_bytes": (_i32, [_vp, _pi64]),
    "cx_state_export": (_i32, [_vp, _vp, C.c_int64]),
    "cx_state_import": (_i32, [_vp, _vp, C.c_int64]),
    "cx_profile_enable": (_i32, [_vp, _i32]),
    "cx_profile_read": (_i32, [_vp, _i32, _pd, _pi64]),
    "cx_kernel_name": (C.c_char_p, [_i32]),
}

def item_range(lo: int, hi: int) -> int:
    """CX_ITEM_RANGE(lo, hi): the 1-based inclusive range of a ProductOfMessages item, as it travels in cx_item.factor_id"""
    return (int(lo) << 32) | int(hi)


_lib = None


class CortexHipError(RuntimeError):
    """Non-zero status from the C ABI (what a Julia shim turns into `error(cx_last_error())`)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"cortex_hip status {code}: {message}")
        self.code = code
        self.message = message


def load():
    """dlopen libcortex_hip.so and type every exported symbol.  Raises if the library is absent:
    the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m cortex.jl_amd.build` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for the device path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.cx_version() != ABI_VERSION:
        raise ImportError(f"libcortex_hip.so ABI {lib.cx_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib
