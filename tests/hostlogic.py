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
        _lib.cxh_plan64_create.argtypes = [C.c_int32, C.c_int64, C.c_int64] + [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_int64, C.c_char_p, C.c_int32]
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

    def __init__(self, d, link_pos, frm, to, tab_fwd, tab_bwd, head_fwd, head_bwd, side, K0=0, fan=4, lanes=1024):
        L = lib()
        a32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
        a8 = lambda x: np.ascontiguousarray(x, dtype=np.uint8)
        link_pos, frm, to, tab_fwd, tab_bwd, side = map(a32, (link_pos, frm, to, tab_fwd, tab_bwd, side))
        head_fwd, head_bwd = a8(head_fwd), a8(head_bwd)
        npos = side.shape[0]
        err = C.create_string_buffer(512)
        p = L.cxh_plan64_create(d, npos, len(link_pos), link_pos.ctypes.data, frm.ctypes.data, to.ctypes.data, tab_fwd.ctypes.data,
                                tab_bwd.ctypes.data, head_fwd.ctypes.data, head_bwd.ctypes.data, side.ctypes.data, K0, fan, lanes, err, 512)
        if not p:
            raise RuntimeError(err.value.decode())
        try:
            info = lambda w: int(L.cxh_plan64_info(p, w))
            self.n_pot, self.n_ent, nch, nst, ncl, nwl, self.msg, self.pot, self.K0, self.levels, self.n_compositions, self.n_rules = (info(w) for w in range(12))
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
