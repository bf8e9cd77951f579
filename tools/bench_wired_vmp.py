#!/usr/bin/env python3
"""The variational state-space model of the reference's tests as a WIRING under CX_SCHED_REFERENCE (cx_graph_wire) next to the fused family
handle (cx_update_marginals): ms per update_marginals! call in the steady state, the plan each call replays, and what the host pays once
(building the triples, wiring, the first call of each kind = the scheduler + levelling + upload).  One JSON line per n."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402


def timed(dev, fn, reps):
    dev.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    dev.sync()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[1000, 100_000, 1_000_000])
    ap.add_argument("--kind", default="structured")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    for n in a.n:
        model = cx.synth.vmp_ssm(n, seed=4)
        nf = len(model.factor_ids)
        row = {"n": n, "kind": a.kind}
        t0 = time.perf_counter()
        if a.kind == "structured":
            t = cx.wiring.structured(model.edge_var, model.edge_fac, model.edge_role, clustered_factors=model.factor_ids[model.n:])
        else:
            t = cx.wiring.mean_field(model.edge_var, model.edge_fac, model.edge_role)
        row["triples"] = int(len(t.flags)); row["build_triples_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32), np.zeros(nf), edge_role=model.edge_role)
        t0 = time.perf_counter()
        dev.graph_wire(t.signals, t.dependencies, t.flags)
        row["graph_wire_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        dev.set_marginals([model.ssnoise, model.obsnoise], L.FORM_GAMMA, [1.0, 1.0, 1.0, 1.0])
        dev.set_marginals(model.x_ids, L.FORM_MEAN_PRECISION, np.tile([0.0, 1.0], model.n))
        dev.set_marginals(model.y_ids, L.FORM_POINT, model.data_y)
        fused = cx.DeviceGraph(family=L.FAMILY_VMP_STRUCTURED if a.kind == "structured" else L.FAMILY_VMP_MEAN_FIELD, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_vmp_into_device(model, fused)
        calls = {"states": model.x_ids, "ssnoise": [model.ssnoise], "obsnoise": [model.obsnoise]}
        first = {}
        for it in range(3):      # the first iterations run the scheduler: the readiness state settles into its cycle
            for name, ids in calls.items():
                t0 = time.perf_counter()
                dev.sweep_for(ids); dev.sync()
                if it == 0:
                    first[name] = round((time.perf_counter() - t0) * 1e3, 1)
                fused.update_marginals(ids)
        row["first_call_ms"] = first
        a_, b_ = dev.get_marginals([model.ssnoise, model.obsnoise]), fused.get_marginals([model.ssnoise, model.obsnoise])
        xa, xb = dev.get_marginals(model.x_ids), fused.get_marginals(model.x_ids)
        row["parity_vs_fused_family"] = {"precisions_max_rel": float(np.max(np.abs(a_ - b_) / np.abs(b_))), "state_means_max_abs": float(np.max(np.abs(xa[:, 0] - xb[:, 0])))}
        steady, plans, fz = {}, {}, {}
        # the steady state: one iteration = the three calls; time each call kind over reps iterations by timing whole iterations and single kinds
        def iteration():
            for ids in calls.values():
                dev.sweep_for(ids)
        def iteration_fused():
            fused.update_marginals(L.VMP_ALL_NORMAL); fused.update_marginals([model.ssnoise]); fused.update_marginals([model.obsnoise])
        before = dev.ref_plan_stats()
        row["iteration_ms"] = round(timed(dev, iteration, a.reps), 4)
        after = dev.ref_plan_stats()
        row["iteration_ms_fused_family"] = round(timed(fused, iteration_fused, a.reps), 4)
        row["plan_hits_during_timing"] = after["hits"] - before["hits"]; row["plan_misses_during_timing"] = after["misses"] - before["misses"]
        for name, ids in calls.items():
            for other, oids in calls.items():      # bring the state to where this call kind starts
                if other == name:
                    break
                dev.sweep_for(oids)
            dev.sync(); t0 = time.perf_counter(); dev.sweep_for(ids); dev.sync()
            steady[name] = round((time.perf_counter() - t0) * 1e3, 4)
            st = dev.ref_plan_stats()
            plans[name] = {k: st[k] for k in ("stages", "launches", "executions", "messages", "rounds")}
            for other in list(calls)[list(calls).index(name) + 1:]:
                dev.sweep_for(calls[other])
        row["call_ms"] = steady; row["plans"] = plans
        print(json.dumps(row), flush=True)
        dev.close(); fused.close()


if __name__ == "__main__":
    main()
