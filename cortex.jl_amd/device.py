"""DeviceGraph — thin Python owner of a `cx_handle` (include/cortex_hip.h).

This is plumbing over the C ABI, not an algorithm: every method is one ABI call.  It is what the
host-side processor (hip_processor.py) and bench.py drive."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class DeviceGraph:
    def __init__(self, device: int = 0, dim: int = 1, schedule: int = L.SCHED_FUSED, marginals_in_sweep: bool = True,
                 materialize_messages_to_factor: bool = False, family: int = L.FAMILY_GAUSSIAN):
        self.lib = L.load()
        self._batch_raw = None
        cfg = L.Config(C.sizeof(L.Config), device, dim, schedule, int(marginals_in_sweep),      # True/1: every sweep; 2: on demand (chain scan, dim 2..4)
                       int(materialize_messages_to_factor), int(family), 0)
        h = C.c_void_p()
        rc = self.lib.cx_create(C.byref(cfg), C.byref(h))
        if rc != L.OK:
            raise L.CortexHipError(rc, self.lib.cx_last_error(None).decode())
        self.h = h
        self.dim = dim
        self.schedule = schedule

    # -- status handling ------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != L.OK:
            raise L.CortexHipError(rc, self.lib.cx_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.cx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- graph ----------------------------------------------------------------------------------
    def graph_create(self, edge_var, edge_fac, factor_ids, factor_kind, factor_params, edge_role=None):
        edge_var, edge_fac, factor_ids = _i64(edge_var), _i64(edge_fac), _i64(factor_ids)
        factor_kind = np.ascontiguousarray(factor_kind, dtype=np.int32)
        fp = np.zeros((len(factor_ids), L.NPARAM), dtype=np.float64)
        factor_params = np.asarray(factor_params, dtype=np.float64)
        if factor_params.ndim == 1:
            fp[:, 0] = factor_params
        else:
            fp[:, : factor_params.shape[1]] = factor_params
        role = None
        if edge_role is not None:
            role_arr = np.ascontiguousarray(edge_role, dtype=np.int32)
            role = _p(role_arr, C.c_int32)
        self._check(self.lib.cx_graph_create(self.h, len(edge_var), _p(edge_var, C.c_int64), _p(edge_fac, C.c_int64),
                                             role, len(factor_ids), _p(factor_ids, C.c_int64),
                                             _p(factor_kind, C.c_int32), _p(fp, C.c_double)))

    def set_factor_coefficients(self, variable_ids, factor_ids, a):
        """cx_set_factor_coefficients: a_i of the ROLE_IN edges of CX_FACTOR_GAUSS_LINEAR_N factors (default 1)"""
        v, f, a = _i64(np.atleast_1d(variable_ids)), _i64(np.atleast_1d(factor_ids)), _f64(np.atleast_1d(a))
        self._check(self.lib.cx_set_factor_coefficients(self.h, len(v), _p(v, C.c_int64), _p(f, C.c_int64), _p(a, C.c_double)))

    def set_factor_matrices(self, parameter_set: int, A, Q):
        A, Q = _f64(A), _f64(Q)
        if A.shape != (self.dim, self.dim) or Q.shape != (self.dim, self.dim):
            raise ValueError(f"A and Q must be {self.dim}x{self.dim}")
        self._check(self.lib.cx_set_factor_matrices(self.h, int(parameter_set), _p(A, C.c_double), _p(Q, C.c_double)))

    def set_factor_edge_sets(self, variable_ids, factor_ids, parameter_sets):
        """cx_set_factor_edge_sets: dim 2..4, the A_i of ROLE_IN edges of CX_FACTOR_GAUSS_LINEAR_N factors by parameter set"""
        v, f, s = _i64(np.atleast_1d(variable_ids)), _i64(np.atleast_1d(factor_ids)), _i64(np.atleast_1d(parameter_sets))
        self._check(self.lib.cx_set_factor_edge_sets(self.h, len(v), _p(v, C.c_int64), _p(f, C.c_int64), _p(s, C.c_int64)))

    def stats(self) -> dict:
        s = L.Stats()
        self._check(self.lib.cx_graph_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.Stats._fields_}

    def tree_plan_stats(self):
        """cx_tree_plan_stats: the stages of the tree schedule's last sweep (zeros before it and for other schedules)"""
        out = (C.c_int64 * 8)()
        self._check(self.lib.cx_tree_plan_stats(self.h, out))
        keys = ("depth", "stages", "items", "kary_entries", "components", "messages_up", "messages_down", "marginals")
        return dict(zip(keys, [int(x) for x in out]))

    def tree_heavy_path_stats(self):
        """cx_tree_heavy_path_stats: zeros unless the tree schedule's last sweep ran over heavy paths"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.cx_tree_heavy_path_stats(self.h, out))
        return dict(zip(("light_depths", "paths", "single_variables", "launches"), [int(x) for x in out]))

    def chain_plan_stats(self):
        """cx_chain_plan_stats: the composition / walk plan of the dim 64 chain-scan schedule (zeros for other handles)"""
        out = (C.c_int64 * 8)()
        self._check(self.lib.cx_chain_plan_stats(self.h, out))
        keys = ("links_per_block", "fan", "levels", "potentials", "compositions", "rules", "launches", "device_bytes")
        return dict(zip(keys, (int(x) for x in out)))

    def chain_scan_stats(self):
        """cx_chain_scan_stats: the scalar chain scan as one launch — state (1 ready, 0 not prepared, -1 off) and how many such launches ran"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.cx_chain_scan_stats(self.h, out))
        return {"state": int(out[0]), "launches": int(out[1]), "halo_batch_graph_launches": int(out[2])}

    def edge_index(self, variable_ids, factor_ids):
        v, f = _i64(np.atleast_1d(variable_ids)), _i64(np.atleast_1d(factor_ids))
        out = np.zeros(len(v), dtype=np.int64)
        self._check(self.lib.cx_edge_index(self.h, len(v), _p(v, C.c_int64), _p(f, C.c_int64), _p(out, C.c_int64)))
        return out

    # -- data -----------------------------------------------------------------------------------
    def set_messages(self, variable_ids, factor_ids, direction: int, form: int, payload):
        v, f = _i64(np.atleast_1d(variable_ids)), _i64(np.atleast_1d(factor_ids))
        p = _f64(payload)
        need = len(v) * self.lib.cx_payload_doubles(self.dim, form)
        if p.size != need:
            raise ValueError(f"payload has {p.size} doubles, expected {need}")
        self._check(self.lib.cx_set_messages(self.h, len(v), _p(v, C.c_int64), _p(f, C.c_int64), direction, form,
                                             _p(p, C.c_double)))

    def get_messages(self, variable_ids, factor_ids, direction: int, form: int = L.FORM_MOMENT):
        v, f = _i64(np.atleast_1d(variable_ids)), _i64(np.atleast_1d(factor_ids))
        out = np.zeros((len(v), self.lib.cx_payload_doubles(self.dim, L.FORM_MOMENT)), dtype=np.float64)
        self._check(self.lib.cx_get_messages(self.h, len(v), _p(v, C.c_int64), _p(f, C.c_int64), direction, form,
                                             _p(out, C.c_double)))
        return out

    def seed_messages(self, direction: int, mean: float, variance: float):
        self._check(self.lib.cx_seed_messages(self.h, direction, float(mean), float(variance)))

    def get_marginals(self, variable_ids):
        v = _i64(np.atleast_1d(variable_ids))
        out = np.zeros((len(v), self.lib.cx_payload_doubles(self.dim, L.FORM_MOMENT)), dtype=np.float64)
        self._check(self.lib.cx_get_marginals(self.h, len(v), _p(v, C.c_int64), _p(out, C.c_double)))
        return out

    # -- compute --------------------------------------------------------------------------------
    def update_batch(self, kinds, variable_ids, factor_ids):
        n = len(kinds)
        items = (L.Item * n)()
        for i in range(n):
            items[i].kind = int(kinds[i])
            items[i].variable_id = int(variable_ids[i])
            items[i].factor_id = int(factor_ids[i])
        self._check(self.lib.cx_update_batch(self.h, items, n))

    def update_batch_packed(self, records: bytes, n: int):
        """cx_update_batch_async (small batches return when their launch is queued) with the items already packed (`struct.pack("<iiqq", kind, 0, variable_id, factor_id)` each, concatenated):
        a scheduler that launches thousands of small batches packs every signal's record once"""
        if self._batch_raw is None:
            proto = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_char_p, C.c_int64)
            self._batch_raw = proto(("cx_update_batch_async", self.lib))
        rc = self._batch_raw(self.h, records, n)
        if rc != L.OK:
            self._check(rc)

    def get_products(self, variable_ids, range_lo, range_hi, form: int = L.FORM_MOMENT):
        """stored ProductOfMessages(variable, lo:hi) values, [n, 2] (dim > 1: [n, d + d * d], as messages)"""
        v = _i64(np.atleast_1d(variable_ids))
        lo = np.ascontiguousarray(np.atleast_1d(range_lo), dtype=np.int32)
        hi = np.ascontiguousarray(np.atleast_1d(range_hi), dtype=np.int32)
        out = np.zeros((len(v), self.lib.cx_payload_doubles(self.dim, L.FORM_MOMENT) if self.dim > 1 else 2), dtype=np.float64)
        self._check(self.lib.cx_get_products(self.h, len(v), _p(v, C.c_int64), _p(lo, C.c_int32), _p(hi, C.c_int32), form,
                                             _p(out, C.c_double)))
        return out

    def get_joint_marginals(self, factor_ids):
        """stored JointMarginal(factor) values: (mean [n, 2], covariance [n, 2, 2]), variables in ascending id order"""
        f = _i64(np.atleast_1d(factor_ids))
        out = np.zeros((len(f), 6), dtype=np.float64)
        self._check(self.lib.cx_get_joint_marginals(self.h, len(f), _p(f, C.c_int64), _p(out, C.c_double)))
        return out[:, :2], out[:, 2:].reshape(-1, 2, 2)

    def sweep(self, n: int = 1):
        self._check(self.lib.cx_sweep(self.h, int(n)))

    def set_damping(self, lam: float):
        """cx_set_damping: new = (1 - lam) rule + lam old for every factor→variable message of a fused / flooding sweep"""
        self._check(self.lib.cx_set_damping(self.h, float(lam)))

    def sweep_for(self, variable_ids):
        """cx_sweep_for: ONE update_marginals!(engine, variable_ids) under CX_SCHED_REFERENCE, the ids in the caller's order"""
        v = _i64(np.atleast_1d(variable_ids))
        self._check(self.lib.cx_sweep_for(self.h, len(v), _p(v, C.c_int64)))

    def graph_wire(self, signals, dependencies, flags):
        """cx_graph_wire: add_dependency!(signal, dependency; flags) triple by triple; a signal is (kind, variable_id, factor_id)"""
        item = np.dtype([("kind", "<i4"), ("reserved", "<i4"), ("variable_id", "<i8"), ("factor_id", "<i8")])      # cx_item
        assert item.itemsize == C.sizeof(L.Item)

        def items(rows):
            rows = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
            out = np.zeros(max(len(rows), 1), dtype=item)
            out["kind"][:len(rows)] = rows[:, 0]; out["variable_id"][:len(rows)] = rows[:, 1]; out["factor_id"][:len(rows)] = rows[:, 2]
            return out

        sig, dep = items(signals), items(dependencies)
        fl = np.ascontiguousarray(flags, dtype=np.int32)
        n = len(fl)
        if len(np.asarray(signals).reshape(-1, 3)) != n or len(np.asarray(dependencies).reshape(-1, 3)) != n:
            raise ValueError("graph_wire: signals, dependencies and flags must have one row per triple")
        self._check(self.lib.cx_graph_wire(self.h, n, sig.ctypes.data_as(C.POINTER(L.Item)), dep.ctypes.data_as(C.POINTER(L.Item)), _p(fl, C.c_int32) if n else None))

    def cluster_stats(self) -> dict:
        """cx_cluster_stats: the XCD-resident cluster — state (1 ready, 0 not prepared, -1 off), workgroups per launch, calls that were
        finished on plain launches after a barrier of the cluster timed out, whether the last reference-order call ran on it"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.cx_cluster_stats(self.h, out))
        return {"state": int(out[0]), "workgroups": int(out[1]), "recovered_calls": int(out[2]), "last_reference_call": bool(out[3])}

    def ref_plan_stats(self) -> dict:
        out = (C.c_int64 * 8)()
        self._check(self.lib.cx_ref_plan_stats(self.h, out))
        keys = ("stages", "launches", "executions", "messages", "rounds", "plans", "hits", "misses")
        return dict(zip(keys, (int(x) for x in out)))

    def ref_trace(self):
        """cx_ref_trace: the executions of the last reference-order call as (kind, variable_id, factor_id, lo, hi) tuples"""
        n = C.c_int64()
        self._check(self.lib.cx_ref_trace(self.h, 0, None, C.byref(n)))
        items = (L.Item * max(1, n.value))()
        self._check(self.lib.cx_ref_trace(self.h, n.value, items, C.byref(n)))
        out = []
        for it in items[:n.value]:
            if it.kind == L.ITEM_PRODUCT_OF_MESSAGES:
                out.append((it.kind, int(it.variable_id), 0, int(it.factor_id) >> 32, int(it.factor_id) & 0xffffffff))
            else:
                out.append((it.kind, int(it.variable_id), int(it.factor_id), 0, 0))
        return out

    def sweep_begin(self):
        self._check(self.lib.cx_sweep_begin(self.h))

    def sweep_main(self):
        self._check(self.lib.cx_sweep_main(self.h))

    def sweep_end(self):
        self._check(self.lib.cx_sweep_end(self.h))

    def sync(self):
        self._check(self.lib.cx_sync(self.h))

    def set_stream(self, stream_ptr: Optional[int]):
        self._check(self.lib.cx_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def residual(self) -> float:
        out = C.c_double()
        self._check(self.lib.cx_residual(self.h, C.byref(out)))
        return out.value

    def message_health(self):
        """cx_message_health: the stored factor→variable messages into non-observed variables, counted on the device by state"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.cx_message_health(self.h, out))
        return dict(zip(("defined", "undefined", "negative_precision", "non_finite"), [int(x) for x in out]))

    # -- halo -----------------------------------------------------------------------------------
    def halo_configure(self, send_var, send_fac, recv_var, recv_fac):
        sv, sf, rv, rf = _i64(send_var), _i64(send_fac), _i64(recv_var), _i64(recv_fac)
        self._check(self.lib.cx_halo_configure(self.h, len(sv), _p(sv, C.c_int64), _p(sf, C.c_int64), len(rv),
                                               _p(rv, C.c_int64), _p(rf, C.c_int64)))

    def sweep_until(self, tol: float, max_sweeps: int, check_every: int = 10):
        """(sweeps run, last residual): cx_sweep_until"""
        n, r = C.c_int32(), C.c_double()
        self._check(self.lib.cx_sweep_until(self.h, float(tol), int(max_sweeps), int(check_every), C.byref(n), C.byref(r)))
        return n.value, r.value

    def halo_configure_state(self, send_var, send_fac, recv_var, recv_fac):
        """Deep halo: the lists name factor→variable messages of redundant variables (cx_halo_configure_state)."""
        sv, sf, rv, rf = _i64(send_var), _i64(send_fac), _i64(recv_var), _i64(recv_fac)
        self._check(self.lib.cx_halo_configure_state(self.h, len(sv), _p(sv, C.c_int64), _p(sf, C.c_int64), len(rv),
                                                     _p(rv, C.c_int64), _p(rf, C.c_int64)))

    def halo_set_layers(self, variable_ids, layers, depth: int):
        v = _i64(np.atleast_1d(variable_ids))
        lay = np.ascontiguousarray(np.atleast_1d(layers), dtype=np.int32)
        self._check(self.lib.cx_halo_set_layers(self.h, len(v), _p(v, C.c_int64), _p(lay, C.c_int32), int(depth)))

    def halo_state_pack(self):
        self._check(self.lib.cx_halo_state_pack(self.h))

    def halo_state_unpack(self):
        self._check(self.lib.cx_halo_state_unpack(self.h))

    def halo_state_exchange(self):
        self._check(self.lib.cx_halo_state_exchange(self.h))

    def halo_exchange_sweep(self, n: int):
        """the exchange and n sweeps, the exchange overlapped with the owned part of the first sweep"""
        self._check(self.lib.cx_halo_exchange_sweep(self.h, int(n)))

    @property
    def halo_doubles(self) -> int:
        """doubles per message in the halo buffers (the storage form)"""
        d = self.dim
        return 2 if d == 1 else (d + d * d if d == 64 else d + d * (d + 1) // 2)

    def chain_block_maps(self):
        """cx_chain_block_maps: (forward map [6], backward map [6], side of the first variable [2], side of the last [2],
        first variable id, last variable id, links)"""
        d = self.dim
        nd, ns = (6, 2) if d == 1 else (d * (d + 1) + d * d + 2 * d, d + d * (d + 1) // 2)     # dim > 1: P | B | C | h | c;  eta | Lambda packed
        f, b, sf, sl = np.zeros(nd), np.zeros(nd), np.zeros(ns), np.zeros(ns)
        v0, v1, nl = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.cx_chain_block_maps(self.h, _p(f, C.c_double), _p(b, C.c_double), _p(sf, C.c_double), _p(sl, C.c_double),
                                                 C.byref(v0), C.byref(v1), C.byref(nl)))
        return f, b, sf, sl, v0.value, v1.value, nl.value

    def halo_buffers(self):
        sp, rp, sb, rb = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
        self._check(self.lib.cx_halo_buffers(self.h, C.byref(sp), C.byref(sb), C.byref(rp), C.byref(rb)))
        return (sp.value or 0, sb.value), (rp.value or 0, rb.value)

    def halo_set_buffers(self, send_ptr: int, recv_ptr: int):
        self._check(self.lib.cx_halo_set_buffers(self.h, C.c_void_p(send_ptr or 0), C.c_void_p(recv_ptr or 0)))

    # -- RCCL exchange issued by the library -------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.cx_comm_unique_id(buf)
        if rc != L.OK:
            raise L.CortexHipError(rc, self.lib.cx_last_error(None).decode())
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        self._check(self.lib.cx_comm_init(self.h, world, rank, C.c_char_p(unique_id)))

    # -- deep-halo exchange through IPC-mapped receive areas (cx_api_ipc.hip) -----------------------
    def halo_ipc_alloc(self):
        """-> (64-byte IPC handle, device address of the block, bytes of one receive area)"""
        buf = C.create_string_buffer(64)
        base, area = C.c_void_p(), C.c_int64()
        self._check(self.lib.cx_halo_ipc_alloc(self.h, buf, C.byref(base), C.byref(area)))
        return buf.raw, int(base.value), int(area.value)

    def halo_ipc_connect(self, peer_index: int, remote_entry: int, remote_recv_off: int, remote_area_bytes: int, handle: bytes = None,
                         same_process_base: int = None):
        assert (handle is None) != (same_process_base is None)
        self._check(self.lib.cx_halo_ipc_connect(self.h, int(peer_index), C.c_char_p(handle) if handle is not None else None,
                                                 C.c_void_p(same_process_base) if same_process_base is not None else None,
                                                 int(remote_entry), int(remote_recv_off), int(remote_area_bytes)))

    def halo_ipc_exchange(self):
        self._check(self.lib.cx_halo_ipc_exchange(self.h))

    def halo_ipc_exchange_sweep(self, n: int):
        """the exchange around the owned part of the first of n sweeps"""
        self._check(self.lib.cx_halo_ipc_exchange_sweep(self.h, int(n)))

    def halo_ipc_batch(self, n: int):
        """cx_halo_ipc_batch: n sweeps with the next exchange pushed inside the last one (bit-identical to exchange + sweep(n))"""
        self._check(self.lib.cx_halo_ipc_batch(self.h, int(n)))

    def halo_ipc_set_fused(self, on: bool):
        """cx_halo_ipc_set_fused: every neighbour pushes from another device -> one launch per exchange"""
        self._check(self.lib.cx_halo_ipc_set_fused(self.h, 1 if on else 0))

    def halo_ipc_push(self):
        self._check(self.lib.cx_halo_ipc_push(self.h))

    def halo_ipc_unpack(self):
        self._check(self.lib.cx_halo_ipc_unpack(self.h))

    def halo_ipc_status(self):
        """synchronises -> (timed_out, exchanges)"""
        t, n = C.c_int32(), C.c_int64()
        self._check(self.lib.cx_halo_ipc_status(self.h, C.byref(t), C.byref(n)))
        return bool(t.value), int(n.value)

    def halo_ipc_set_timeout(self, seconds: float):
        self._check(self.lib.cx_halo_ipc_set_timeout(self.h, float(seconds)))

    def halo_peers(self, peers):
        """peers: iterable of (rank, send_offset, send_count, recv_offset, recv_count), in messages."""
        peers = list(peers)
        r = np.ascontiguousarray([p[0] for p in peers], dtype=np.int32)
        cols = [np.ascontiguousarray([p[k] for p in peers], dtype=np.int64) for k in (1, 2, 3, 4)]
        self._check(self.lib.cx_halo_peers(self.h, len(peers), _p(r, C.c_int32), *[_p(c, C.c_int64) for c in cols]))

    def sweep_exchange(self, n: int = 1):
        self._check(self.lib.cx_sweep_exchange(self.h, int(n)))

    # -- variational families (cx_set_marginals / cx_update_marginals) ----------------------------
    def set_marginals(self, variable_ids, form: int, payload):
        v = _i64(np.atleast_1d(variable_ids))
        p = _f64(payload)
        need = len(v) * (1 if form == L.FORM_POINT else 2)
        if p.size != need:
            raise ValueError(f"payload has {p.size} doubles, expected {need}")
        self._check(self.lib.cx_set_marginals(self.h, len(v), _p(v, C.c_int64), form, _p(p, C.c_double)))

    def update_marginals(self, variable_ids):
        """One `update_marginals!(engine, ids)`; `variable_ids` may be L.VMP_ALL_NORMAL / L.VMP_ALL_PRECISION."""
        if isinstance(variable_ids, (int, np.integer)) and variable_ids < 0:
            self._check(self.lib.cx_update_marginals(self.h, int(variable_ids), None))
            return
        v = _i64(np.atleast_1d(variable_ids))
        self._check(self.lib.cx_update_marginals(self.h, len(v), _p(v, C.c_int64)))

    # -- checkpoint (cx_state_*) -----------------------------------------------------------------
    def export_state(self) -> np.ndarray:
        """The handle's mutable state (messages, marginals, observed flags, sweep counter) as a uint8 array."""
        n = C.c_int64()
        self._check(self.lib.cx_state_bytes(self.h, C.byref(n)))
        blob = np.empty(n.value, dtype=np.uint8)
        self._check(self.lib.cx_state_export(self.h, blob.ctypes.data_as(C.c_void_p), n.value))
        return blob

    def import_state(self, blob) -> None:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self.lib.cx_state_import(self.h, blob.ctypes.data_as(C.c_void_p), blob.size))

    def save_state(self, path: str) -> None:
        self.export_state().tofile(path)

    def load_state(self, path: str) -> None:
        self.import_state(np.fromfile(path, dtype=np.uint8))

    # -- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """True/1: hipEvents around every launch; n > 1: around every n-th launch; False/0: off."""
        self._check(self.lib.cx_profile_enable(self.h, int(on)))

    def profile_read(self, kernel: int):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.cx_profile_read(self.h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_name(self, kernel: int) -> str:
        return self.lib.cx_kernel_name(kernel).decode()
