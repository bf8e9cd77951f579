"""ctypes view of cortex.jl_amd/libcortex_hostlogic.so — the product's GPU-free host logic compiled for the CPU
(cortex.jl_amd/csrc/cx_hostlogic.cpp; `CXH_LIB` selects another build, e.g. the address/UB-sanitizer one)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from cortex.jl_amd import build as B

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("CXH_LIB") or B.build_hostlogic()
        _lib = C.CDLL(path)
        _lib.cxh_plan64_create.restype = C.c_void_p
        _lib.cxh_plan64_create.argtypes = [C.c_int32, C.c_int64, C.c_int64] + [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_char_p, C.c_int32]
        _lib.cxh_plan64_destroy.argtypes = [C.c_void_p]
        _lib.cxh_plan64_info.restype = C.c_int64
        _lib.cxh_plan64_info.argtypes = [C.c_void_p, C.c_int32]
        _lib.cxh_plan64_jobs.restype = C.c_int64
        _lib.cxh_plan64_jobs.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _lib.cxh_plan64_records.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    return _lib


SPACES = ("zero", "f2v", "ptab", "btab", "pot", "ent")


class Plan64:
    """the plan of cx_chain64_plan.h as numpy arrays"""

    def __init__(self, d, link_pos, frm, to, tab_fwd, tab_bwd, head_fwd, head_bwd, side, K0=0, fan=4, lanes=1024, root=False):
        L = lib()
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        a8 = lambda x: np.ascontiguousarray(x, dtype=np.uint8)
        link_pos, frm, to, tab_fwd, tab_bwd, side = map(a32, (link_pos, frm, to, tab_fwd, tab_bwd, side))
        head_fwd, head_bwd = a8(head_fwd), a8(head_bwd)
        npos = side.shape[0]
        err = C.create_string_buffer(512)
        p = L.cxh_plan64_create(d, npos, len(link_pos), link_pos.ctypes.data, frm.ctypes.data, to.ctypes.data, tab_fwd.ctypes.data,
                                tab_bwd.ctypes.data, head_fwd.ctypes.data, head_bwd.ctypes.data, side.ctypes.data, K0, fan, lanes, 1 if root else 0, err, 512)
        if not p:
            raise RuntimeError(err.value.decode())
        try:
            info = lambda w: int(L.cxh_plan64_info(p, w))
            self.n_pot, self.n_ent, nch, nst, ncl, nwl, self.msg, self.pot, self.K0, self.levels, self.n_compositions, self.n_rules = (info(w) for w in range(12))
            self.root_pot = [info(100 + i) for i in range(info(12))]
            self.children = np.zeros((nch, 10), dtype=np.int64)
            self.steps = np.zeros((nst, 10), dtype=np.int64)
            if nch:
                L.cxh_plan64_records(p, 0, self.children.ctypes.data)
            if nst:
                L.cxh_plan64_records(p, 1, self.steps.ctypes.data)

            def jobs(kind, n):
                out = []
                for i in range(n):
                    k = int(L.cxh_plan64_jobs(p, kind, i, None))
                    a = np.zeros((k, 3), dtype=np.int64)
                    if k:
                        L.cxh_plan64_jobs(p, kind, i, a.ctypes.data)
                    out.append(a)
                return out

            self.compose_launches, self.walk_launches = jobs(0, ncl), jobs(1, nwl)
        finally:
            L.cxh_plan64_destroy(p)

    @staticmethod
    def split(handle):
        return SPACES[int(handle) >> 56], int(handle) & ((1 << 56) - 1)


# ---- cx_flatten.h / cx_chains.h ------------------------------------------------------------------------------------------------------
_ARR = {"var_ids": 0, "var_off": 1, "edge_var": 2, "edge_fac_id": 3, "vbase": 4, "vinfo": 5, "slice_off": 6, "partner": 7, "big_vars": 8, "spdir": 9,
        "var_deg": 10, "q": 20, "a": 21, "b": 22, "sq": 23, "sa": 24, "sb": 25, "kary_coef": 26, "kary_qb": 27, "kary_slot": 30, "slot_kary": 31,
        "pos_var": 40, "skip0": 41, "skip1": 42, "link_pos": 43, "from": 44, "to": 45, "head_fwd": 46, "head_bwd": 47, "tab_fwd": 48, "tab_bwd": 49,
        "trim_lo": 50, "trim_hi": 51, "hp_rec": 70, "hp_stage_off": 71, "hp_kary": 72, "hp_kary_off": 73, "hp_pos_var": 74, "hp_skip0": 75, "hp_skip1_up": 76, "hp_skip1_down": 77,
        "hp_link_pos": 78, "hp_from": 79, "hp_to": 80, "hp_head_fwd": 81, "hp_head_bwd": 82, "hp_pos_off": 83, "hp_link_off": 84, "hp_steps": 85,
        "ref_rec": 90, "ref_stage_off": 91, "ref_list": 92, "ref_dep_off": 93, "ref_dep": 94, "ref_dep_inter": 95, "ref_flags": 96, "ref_order": 97, "ref_wide_rec": 98, "ref_wide_off": 99,
        "ref_scans": 100, "ref_sl_lead_dst": 101, "ref_sl_lead_var": 102, "ref_sl_fol_dst": 103, "ref_sl_prec": 104, "ref_sl_src_off": 105, "ref_sl_src": 106, "ref_sl_head": 107,
        "tree_rec": 60, "tree_stage_off": 61, "tree_kary": 62, "tree_kary_off": 63, "partner": 64, "slot_kary_all": 65,
        "kary_slot_all": 66}
_SCA = {"nv": 0, "nf": 1, "ne": 2, "nslots": 3, "nslices": 4, "n_messages_per_sweep": 5, "any_linear": 6, "n_kary": 7, "big_start": 8, "npos_linked": 9,
        "own_slice_lo": 10, "own_slice_hi": 11, "ipc_quiet_lo": 12, "ipc_quiet_hi": 13, "tree_depth": 14, "tree_components": 15, "tree_up": 16,
        "tree_down": 17, "tree_marginals": 18, "hp_levels": 19, "hp_paths": 20, "hp_single": 21, "hp_launches": 22, "hp_marginal_stage": 23, "hp_kary_links": 24}


class FlatGraph:
    """cx_graph_create's flattening (and, on request, the chain decomposition) run on the CPU build of the product's host logic"""

    def __init__(self, edge_var, edge_fac, factor_ids, factor_kind, factor_params, edge_role=None, dim=1, schedule=1, family=0):
        L = lib()
        if not hasattr(L, "_flat_ready"):
            L.cxh_flat_create.restype = C.c_void_p
            L.cxh_flat_create.argtypes = [C.c_int32] * 3 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
            L.cxh_flat_destroy.argtypes = [C.c_void_p]
            L.cxh_flat_clamp.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            L.cxh_flat_chains.restype = C.c_int32
            L.cxh_flat_chains.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_flat_array.restype = C.c_int64
            L.cxh_flat_array.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
            L.cxh_flat_scalar.restype = C.c_int64
            L.cxh_flat_scalar.argtypes = [C.c_void_p, C.c_int32]
            L.cxh_flat_tree.restype = C.c_int32
            L.cxh_flat_tree.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_flat_tree_hp.restype = C.c_int32
            L.cxh_flat_tree_hp.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_flat_halo.restype = C.c_int32
            L.cxh_flat_halo.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
            L.cxh_ref_build.restype = C.c_int32
            L.cxh_ref_build.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_ref_wire.restype = C.c_int32
            L.cxh_ref_wire.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_ref_set.restype = C.c_int32
            L.cxh_ref_set.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
            L.cxh_ref_set_marginals.restype = C.c_int32
            L.cxh_ref_set_marginals.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            L.cxh_ref_update.restype = C.c_int64
            L.cxh_ref_update.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            L.cxh_ref_trace.argtypes = [C.c_void_p, C.c_void_p]
            L.cxh_ref_level.restype = C.c_int32
            L.cxh_ref_level.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
            L.cxh_ref_scalar.restype = C.c_int64
            L.cxh_ref_scalar.argtypes = [C.c_void_p, C.c_int32]
            L._flat_ready = True
        self.L = L
        ev = np.ascontiguousarray(edge_var, dtype=np.int64); ef = np.ascontiguousarray(edge_fac, dtype=np.int64)
        fi = np.ascontiguousarray(factor_ids, dtype=np.int64); fk = np.ascontiguousarray(factor_kind, dtype=np.int32)
        fp = np.zeros((len(fi), 4)); raw = np.asarray(factor_params, dtype=np.float64)
        if raw.ndim == 1:
            fp[:, 0] = raw
        else:
            fp[:, :raw.shape[1]] = raw
        role = None if edge_role is None else np.ascontiguousarray(edge_role, dtype=np.int32)
        st, err = C.c_int32(0), C.create_string_buffer(512)
        self.p = L.cxh_flat_create(dim, schedule, family, len(ev), ev.ctypes.data, ef.ctypes.data, None if role is None else role.ctypes.data,
                                   len(fi), fi.ctypes.data, fk.ctypes.data, fp.ctypes.data, C.byref(st), err, 512)
        self.status, self.error = st.value, err.value.decode()

    def __del__(self):
        if getattr(self, "p", None):
            self.L.cxh_flat_destroy(self.p)
            self.p = None

    def arr(self, name):
        w = _ARR[name]
        n = int(self.L.cxh_flat_array(self.p, w, None))
        out = np.zeros(max(n, 0), dtype=np.float64 if 20 <= w < 30 else np.int64)
        if n > 0:
            self.L.cxh_flat_array(self.p, w, out.ctypes.data)
        return out

    def scalar(self, name):
        return int(self.L.cxh_flat_scalar(self.p, _SCA[name]))

    def clamp(self, variable_ids):
        v = np.ascontiguousarray(variable_ids, dtype=np.int64)
        self.L.cxh_flat_clamp(self.p, len(v), v.ctypes.data)

    def chains(self):
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_flat_chains(self.p, err, 512))
        return rc, err.value.decode()

    def tree(self):
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_flat_tree(self.p, err, 512))
        return rc, err.value.decode()

    def tree_hp(self):
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_flat_tree_hp(self.p, err, 512))
        return rc, err.value.decode()

    def halo(self, layer_var, layer, depth, send_slots):
        v = np.ascontiguousarray(layer_var, dtype=np.int64); l = np.ascontiguousarray(layer, dtype=np.int32)
        ss = np.ascontiguousarray(send_slots, dtype=np.int32)
        return int(self.L.cxh_flat_halo(self.p, len(v), v.ctypes.data, l.ctypes.data, int(depth), len(ss), ss.ctypes.data))

    # ---- CX_SCHED_REFERENCE (cx_refsched.h): wiring, shadow of the readiness state, recorded and levelled calls --------------------
    def ref_build(self):
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_ref_build(self.p, err, 512))
        return rc, err.value.decode()

    def ref_wire(self, signals, dependencies, flags):
        """a user wiring: add_dependency!(signal, dependency; flags) triple by triple; a signal is (kind, variable id, factor id)"""
        a = np.ascontiguousarray(np.asarray(signals, dtype=np.int64).reshape(-1, 3)); b = np.ascontiguousarray(np.asarray(dependencies, dtype=np.int64).reshape(-1, 3))
        fl = np.ascontiguousarray(flags, dtype=np.int32)
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_ref_wire(self.p, len(fl), a.ctypes.data, b.ctypes.data, fl.ctypes.data, err, 512))
        return rc, err.value.decode()

    def ref_set(self, direction, variable_ids, factor_ids):
        v = np.ascontiguousarray(variable_ids, dtype=np.int64); f = np.ascontiguousarray(factor_ids, dtype=np.int64)
        rc = int(self.L.cxh_ref_set(self.p, int(direction), len(v), v.ctypes.data, f.ctypes.data))
        assert rc == 0, rc

    def ref_set_marginals(self, variable_ids):
        v = np.ascontiguousarray(variable_ids, dtype=np.int64)
        rc = int(self.L.cxh_ref_set_marginals(self.p, len(v), v.ctypes.data))
        assert rc == 0, rc

    def ref_update(self, variable_ids):
        """one update_marginals!(ids) on the shadow: rows {kind, variable id, factor id, lo, hi, round} in execution order"""
        v = np.ascontiguousarray(variable_ids, dtype=np.int64)
        n = int(self.L.cxh_ref_update(self.p, len(v), v.ctypes.data))
        assert n >= 0, n
        out = np.zeros((n, 6), dtype=np.int64)
        if n:
            self.L.cxh_ref_trace(self.p, out.ctypes.data)
        return out

    def ref_level(self):
        err = C.create_string_buffer(512)
        rc = int(self.L.cxh_ref_level(self.p, err, 512))
        return rc, err.value.decode()

    def ref_scalar(self, what):
        return int(self.L.cxh_ref_scalar(self.p, {"signals": 0, "dependencies": 1, "products": 2, "hash": 3, "rounds": 4}[what]))
