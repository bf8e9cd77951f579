#!/usr/bin/env python3
"""tools/bench_configs.py — timings of the non-headline configs of BASELINE.json (C2 chain scan, C3 d=4, C5 d=64).
Not the driver's bench (bench.py measures C4); prints one JSON line per config for DESIGN.md / profiles/.
    python tools/bench_configs.py [c2] [c3] [c5] [vmp]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F64_MATRIX_PEAK_TF = 78.6  # f64 matrix peak (AMD public spec)
F64_MATRIX_SUSTAINED_TF = 77.3   # what a bare v_mfma_f64_16x16x4_f64 loop sustains on this chip with the same number of waves on every
                                 # SIMD (tools/lab/mfma64_peak.hip, profiles/r02_f64_mfma_sustained.txt): the spec figure is reachable
MFMA_PER_MESSAGE = {"wave": 384, "workgroup": 584}    # matrix instructions per factor→variable message of the two d = 64 kernel forms


def counter_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the newest profiles/*_configs_rocprof.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate passes, FETCH x2 on gfx950: tools/profile_configs.sh); None when no summary is on file"""
    import glob
    from importlib import import_module
    sha = import_module("cortex.jl_amd.build").sources_sha16
    subs = [kernel_substr] if isinstance(kernel_substr, str) else list(kernel_substr)
    best, stale = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_configs_rocprof.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_vmp_rocprof.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        hit = {k: v for k, v in d.get("traffic", {}).items() if any(x in k for x in subs)}
        # a figure measured on another version of the kernel is no figure: the summary stores a hash of the kernel's sources
        if not hit:
            continue
        if any(v.get("sources_sha16") != sha(k) for k, v in hit.items()):
            stale = os.path.relpath(f, ROOT)
            continue
        # several kernels (one launch of each per sweep): their sum; `launches_per_sweep` when a kernel runs more than once
        best = (sum(v["hbm_bytes_per_launch"] * v.get("launches_per_sweep", 1) for v in hit.values()), os.path.relpath(f, ROOT))
    if best is None and stale is not None:
        # LOUD: a row of the driver's line is about to lose its counter-based fraction because the kernel changed after the last profile
        print(f"[bench] STALE PROFILE: the counter traffic of {subs} in {stale} was measured on another version of the kernel's sources — "
              f"its roofline row gets frac = null until tools/profile_configs.sh (profile_vmp.sh) is re-run and the summary copied into profiles/",
              file=sys.stderr, flush=True)
    return best


def roofline(bound, achieved, peak, unit, traffic, **extra):
    frac = achieved / peak
    r = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": frac, "traffic": traffic}
    if traffic is None and bound == "hbm":
        # without counter traffic (none on file, or measured on another version of the kernel's sources) only the SURVEY §8d
        # convention is available, which counts bytes these kernels do not move: not a physical fraction, so none is printed
        r["frac"] = None
        r["frac_note"] = "no counter traffic on file for this version of the kernel (tools/profile_configs.sh); the algorithmic convention over-counts its bytes"
    r.update(extra)
    return r


def timed(dev, fn, steps, warmup, reps=None):
    """seconds per call of fn: the MEDIAN of `reps` batches of `steps` calls (3; CX_BENCH_REPS=1 under the profilers, whose summaries count
    launches per iteration).  One batch is not enough in a process that has torn down
    another handle shortly before: one of the first batches of the next handle then holds a stall of ~ 70 ms whatever the kernels are
    (tools/lab/tiles_after_c5.py: 3.4 - 3.8 ms per sweep in ONE batch of 20 around a 0.40 ms kernel, 0.33 before and after; a bare HIP
    program that creates and destroys streams, memory and events does not show it: tools/lab/teardown_stall.hip) — the d = 16 row of
    bench.py came out at 4.3 ms behind the C5 row that way."""
    reps = reps or int(os.environ.get("CX_BENCH_REPS", "3"))
    for _ in range(warmup):
        fn()
    dev.sync()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        dev.sync()
        out.append((time.perf_counter() - t0) / steps)
    return sorted(out)[len(out) // 2]


def c2(check=None):
    """check(dev, model) -> dict: an optional parity callback run on the timed device (bench.py passes one that uses the CPU checker;
    nothing under tools/ reaches into the checker)"""
    T = 250_001
    model = cx.synth.ssm_chain(T, seed=1234)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    dt = timed(dev, lambda: dev.sweep(1), 50, 5)
    parity = check(dev, model) if check else None
    st = dev.stats()
    # one reference update_marginals! on this chain is 5T-4 message computations + T marginals
    # the same chain under the fused flooding schedule: ONE parallel sweep (information moves one step; T sweeps converge)
    fl = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, fl, seed_variance=1e6)
    fl.sweep(3)
    dtf = timed(fl, lambda: fl.sweep(1), 200, 20)
    nf = fl.stats()["n_messages_per_sweep"]
    one = dev.chain_scan_stats()["launches"] > 0      # (round 6) the scan as ONE launch: k_chain_onepass
    tr = counter_traffic(["k_chain_onepass<4, 256, true"] if one else ["k_chain_run_totals", "k_chain_run_apply"])      # (the instance that also writes the marginals: every sweep but a handle's first)   # the steady-state launches of one sweep
    alg = (5 * T - 4) * 32
    achieved = (tr[0] if tr else alg) / dt / 1e9
    return {"config": "C2", "workload": f"scalar chain T={T} ({st['n_edges']} edges), chain-scan schedule: exact forward/backward in one sweep",
            "ms_per_sweep": dt * 1e3, "reference_updates_per_sweep": 5 * T - 4, "updates_per_s": (5 * T - 4) / dt,
            "algorithmic_GBps": alg / dt / 1e9,
            # two launches at their fixed cost: the sweep is LAUNCH-LATENCY bound, far from the HBM roofline it is priced against
            "roofline": roofline("hbm", achieved, HBM_PEAK_GBS, "GB/s", tr[0] if tr else None, limited_by="one launch: the links' dependent loads and the look-back's trips to memory" if one else "launch latency (2 kernels per sweep)",
                                 kernel="k_chain_onepass" if one else "k_chain_run_totals + k_chain_run_apply",
                                 basis="counter traffic of the sweep's kernels / sweep time" if tr else "algorithmic bytes / sweep time",
                                 traffic_source=tr[1] if tr else None, algorithmic_bytes_per_sweep=alg, frac_survey_convention=alg / dt / 1e9 / HBM_PEAK_GBS),
            "flooding": {"ms_per_sweep": dtf * 1e3, "updates_per_sweep": nf, "updates_per_s": nf / dtf, "algorithmic_GBps": nf * 32 / dtf / 1e9},
            **({"parity": parity} if parity else {})}


def mv(d, T, steps):
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(2)
    dev.profile_enable(1)
    dt = timed(dev, lambda: dev.sweep(1), steps, 3)
    ms, n = dev.profile_read(L.KERNEL_FUSED)
    st = dev.stats()
    upd = 2 * (2 * (T - 1))        # per sweep: both directions on every transition edge: v→f and f→v (messages with readers)
    payload = (d + d * d) * 8
    out = {"config": "C3" if d == 4 else "C5", "workload": f"d={d} linear-Gaussian chain T={T} ({st['n_edges']} edges), fused flooding sweep",
           "ms_per_sweep": dt * 1e3, "kernel_ms": ms / max(n, 1), "updates_per_sweep": upd, "updates_per_s": upd / dt,
           "algorithmic_GBps": upd * 2 * payload / dt / 1e9, "payload_bytes": payload}
    kern_s = ms / max(n, 1) / 1e3
    tr = counter_traffic("k_rule64w<2, 4>" if d == 64 else f"k_sweep_mv<{d},")
    alg = upd * 2 * payload
    if d == 64:
        # v_mfma_f64_16x16x4_f64 per factor→variable message, 2*16*16*4 flop each: 384 in the wave-per-message kernel
        # (Cholesky 64, solve 160, Gram 160), 584 in round 1's workgroup form (CX_RULE64=g) for the same arithmetic
        form = "workgroup" if os.environ.get("CX_RULE64", "") == "g" else "wave"
        nmsg = 2 * (T - 1)
        per = MFMA_PER_MESSAGE[form]
        tf = nmsg * per * 2048 / kern_s / 1e12
        out["mfma_TFLOPs"] = tf
        out["roofline"] = roofline("mfma", tf, F64_MATRIX_PEAK_TF, "TFLOP/s", tr[0] if tr else None,
                                   kernel="k_rule64w" if form == "wave" else "k_rule64s",
                                   mfma_per_message=per, messages_per_launch=nmsg, avg_kernel_ms=kern_s * 1e3,
                                   peak_sustained=F64_MATRIX_SUSTAINED_TF, frac_of_sustained=tf / F64_MATRIX_SUSTAINED_TF,
                                   peak_note="78.6 is the spec figure; a bare MFMA loop on this chip sustains 77.3 (profiles/r02_f64_mfma_sustained.txt)",
                                   round1_equivalent_TFLOPs=nmsg * 584 * 2048 / kern_s / 1e12,
                                   round1_equivalent_note="the same messages priced at round 1's 584 matrix instructions each: what the earlier kernel "
                                                          "would have had to sustain for this time",
                                   traffic_source=tr[1] if tr else None, hbm_GBps=(tr[0] / kern_s / 1e9) if tr else None)
    else:
        achieved = (tr[0] if tr else alg) / kern_s / 1e9
        out["roofline"] = roofline("hbm", achieved, HBM_PEAK_GBS, "GB/s", tr[0] if tr else None, kernel=f"k_sweep_mv<{d}>", avg_kernel_ms=kern_s * 1e3,
                                   basis="counter traffic / avg launch duration" if tr else "algorithmic bytes / avg launch duration",
                                   traffic_source=tr[1] if tr else None, algorithmic_bytes_per_launch=alg,
                                   frac_survey_convention=alg / kern_s / 1e9 / HBM_PEAK_GBS,
                                   survey_convention_note="SURVEY §8d convention: 2 x 160 B per update over ALL directed updates; the kernel skips the constant "
                                                         "messages out of observed variables and stores packed-symmetric 112 B, so it moves fewer bytes")
    return out


def mv_tiles(d=16, T=100_000, steps=20, embedded_T=None, native_only=False):
    """(round 6) a d-dimensional chain on the matrix-core path in its NATIVE tile size (d <= 16: one 16 x 16 tile, d <= 32: 2 x 2) beside the
    same kind of model embedded in 4 x 4 tiles (CX_MFMA_DIM=64: the only form until round 5; `embedded_T` states, default T — the embedding
    of T = 10^5 holds 40 GB): ms per fused sweep, ns per message, bytes per slot, MFMA rate.  The wave-per-message rule issues
    4 NT^3 + 2 NT^2 (NT + 1) matrix instructions for NT x NT tiles: 8 at NT = 1, 56 at NT = 2, 384 at NT = 4."""
    nd = 16 if d <= 16 else 32 if d <= 32 else 64
    nt = nd // 16
    mfma = 4 * nt ** 3 + 2 * nt * nt * (nt + 1)
    rows = {}
    for name, env, Tn in (("native", None, T), ("embedded_in_64", "64", embedded_T or T))[:1 if native_only else 2]:
        model = cx.synth.lgssm_chain(Tn, d=d, seed=1234)
        if env:
            os.environ["CX_MFMA_DIM"] = env
        try:
            dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
        finally:
            os.environ.pop("CX_MFMA_DIM", None)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        dev.sweep(2)
        dev.profile_enable(1)
        dt = timed(dev, lambda: dev.sweep(1), steps if name == "native" else max(4, steps // 4), 2)
        ms, n = dev.profile_read(L.KERNEL_FUSED)
        dev.profile_enable(0)
        st = dev.stats()
        rows[name] = {"states": Tn, "ms_per_sweep": dt * 1e3, "kernel_ms": ms / max(n, 1), "ns_per_message": dt * 1e9 / (2 * (Tn - 1)),
                      "device_bytes": st["device_bytes"], "bytes_per_slot": st["device_bytes"] / st["n_slots"]}
        dev.close()
    if native_only:      # (under the profilers: the native kernel's launches alone)
        return {"config": f"d={d} native tiles", "native": rows["native"]}
    # the same two forms under the chain-scan schedule: ONE exact sweep (the smoother at every state)
    for name, env, Tn in (("native", None, T), ("embedded_in_64", "64", embedded_T or T)):
        model = cx.synth.lgssm_chain(Tn, d=d, seed=1234)
        if env:
            os.environ["CX_MFMA_DIM"] = env
        try:
            dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
        finally:
            os.environ.pop("CX_MFMA_DIM", None)
        cx.synth.load_into_device(model, dev)
        dev.sweep(2)
        dt = timed(dev, lambda: dev.sweep(1), 4, 1)
        rows[name]["chain_scan"] = {"ms_per_exact_sweep": dt * 1e3, "ns_per_state": dt * 1e9 / Tn, "plan": dev.chain_plan_stats()}
        dev.close()
    nmsg = 2 * (T - 1)
    kern_s = rows["native"]["kernel_ms"] * 1e-3
    tr = counter_traffic(f"k_rule64w<4, {nt}>") if T == 100_000 else None      # (profiled at this size: tools/profile_configs.sh)
    tf = nmsg * mfma * 2048 / kern_s / 1e12
    payload = (nd + nd * nd) * 8
    return {"config": f"d={d} native tiles", "workload": f"d={d} linear-Gaussian chain T={T}, fused flooding sweep, {nt} x {nt} tiles of 16 (the embedding in 4 x 4 tiles beside it)",
            "ms_per_sweep": rows["native"]["ms_per_sweep"], "native": rows["native"], "embedded_in_64": rows["embedded_in_64"],
            "speedup_over_embedding": rows["embedded_in_64"]["ns_per_message"] / rows["native"]["ns_per_message"],
            "chain_scan_speedup_over_embedding": rows["embedded_in_64"]["chain_scan"]["ns_per_state"] / rows["native"]["chain_scan"]["ns_per_state"],
            "bytes_ratio": rows["native"]["bytes_per_slot"] / rows["embedded_in_64"]["bytes_per_slot"], "bytes_ratio_of_records": (nd + nd * nd) / (64 + 64 * 64),
            "roofline": roofline("hbm", (tr[0] if tr else nmsg * 2 * payload) / kern_s / 1e9, HBM_PEAK_GBS, "GB/s", tr[0] if tr else None, kernel=f"k_rule64w<4, {nt}>", avg_kernel_ms=kern_s * 1e3,
                                 basis=(f"counter bytes per launch ({tr[1]}) / avg launch duration" if tr else f"record bytes ({2 * payload} B per message: read + written) / avg launch duration; no counter traffic on file"),
                                 mfma_per_message=mfma, mfma_TFLOPs=tf, frac_of_f64_matrix_peak=tf / F64_MATRIX_PEAK_TF,
                                 frac_survey_convention=nmsg * 2 * payload / kern_s / 1e9 / HBM_PEAK_GBS,
                                 bound_detail=f"a message of {nt} x {nt} tiles is {payload} B and {mfma} matrix instructions behind {16 * nt} dependent pivot steps: "
                                              "bound by that chain's latency at four waves per SIMD, neither by bandwidth nor by the matrix pipe")}


def mv_scan(d, T, steps, ks=(None,), check=None):
    """the chain-scan schedule for dim 2..4 (cx_mvchain.hip): ONE sweep = the exact smoother = one reference update_marginals!
    (5T-4 message computations + T marginals).  ks: values of CX_MVC_K (links per thread) to time; None = the default."""
    model = cx.synth.lgssm_chain(T, d=d, seed=1234)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    st = dev.stats()
    payload = (d + d * d) * 8
    ref_upd = 5 * T - 4
    out = []
    for k in ks:
        if k is None:
            os.environ.pop("CX_MVC_K", None)
        else:
            os.environ["CX_MVC_K"] = str(k)
        dev.sweep(2)
        dt = timed(dev, lambda: dev.sweep(1), steps, 3)
        # the same sweep when the data changed: the constant messages out of the observed variables and the side sums are recomputed
        # first.  Re-setting ONE datum marks exactly that without moving the whole data set over PCIe.
        def fresh():
            dev.set_messages(model.data_var[:1], model.data_fac[:1], L.TO_FACTOR, L.FORM_POINT, model.data_y[:1])
            dev.sweep(1)
        dt_fresh = timed(dev, fresh, max(steps // 3, 3), 2)
        tr = counter_traffic(["k_mvc_totals", "k_mvc_apply", "k_mvc_marg_out"]) if k is None else None
        parity = check(dev, model) if (check and k is None) else None
        lazy = None
        if k is None:
            # compute_marginals_in_sweep = 2: the sweep leaves the forward and backward sums of the chain, the marginal pass runs before
            # the first reader (what dim 64 always does).  ms_per_sweep above stays the figure with every marginal written every sweep.
            dev2 = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN, marginals_in_sweep=2)
            cx.synth.load_into_device(model, dev2)
            dev2.sweep(2)
            dt2 = timed(dev2, lambda: dev2.sweep(1), steps, 3)
            one = model.var_ids[:1] if hasattr(model, "var_ids") else np.asarray([int(model.edge_var[0])])
            def sweep_and_read():
                dev2.sweep(1)
                dev2.get_marginals(one)
            dt3 = timed(dev2, sweep_and_read, max(steps // 3, 3), 2)
            lazy = {"ms_per_sweep": dt2 * 1e3, "ms_per_sweep_then_one_marginal_read": dt3 * 1e3,
                    "note": "compute_marginals_in_sweep = 2: a sweep = every chain message (forward and backward sums in the walks' order); a "
                            "cx_get_marginals for a few variables forms just those from the sums (k_mvc_marg_gather), one for an eighth of the chain or "
                            "more runs k_mvc_marg_out once (all marginals to moment form, in place) — the second figure is a sweep followed by the read "
                            "of one marginal, host round trip included",
                    **({"parity": check(dev2, model)} if check else {})}
            dev2.close()
        alg = ref_upd * 2 * payload
        achieved = (tr[0] if tr else alg) / dt / 1e9
        out.append({"config": "C3-scan" if d == 4 else f"d{d}-scan", "links_per_thread": k,
                    "workload": f"d={d} linear-Gaussian chain T={T} ({st['n_edges']} edges), chain-scan schedule: exact forward/backward in one sweep",
                    "ms_per_sweep": dt * 1e3, "reference_updates_per_sweep": ref_upd, "updates_per_s": ref_upd / dt,
                    "ms_per_sweep_after_new_inputs": dt_fresh * 1e3,
                    "ms_per_sweep_after_new_inputs_note": "incl. the leaf passes (messages out of observed variables) and the side sums; the upload of new data itself is PCIe",
                    "roofline": roofline("hbm", achieved, HBM_PEAK_GBS, "GB/s", tr[0] if tr else None, kernel="k_mvc_totals + k_mvc_apply + k_mvc_marg_out",
                                         basis="counter traffic of the sweep's three launches / sweep time" if tr else "algorithmic bytes / sweep time",
                                         traffic_source=tr[1] if tr else None, algorithmic_bytes_per_sweep=alg,
                                         frac_survey_convention=alg / dt / 1e9 / HBM_PEAK_GBS),
                    **({"parity": parity} if parity else {}), **({"marginals_on_demand": lazy} if lazy else {})})
    os.environ.pop("CX_MVC_K", None)
    return out


def mv64_scan(T, steps, ks=(None,), check=None):
    """the chain-scan schedule for dim 64 (cx_mv64chain.hip): ONE sweep = the exact smoother = one reference update_marginals!
    (5T-4 message computations).  Matrix work per sweep from the plan: 960 v_mfma_f64_16x16x4_f64 per pairwise composition
    of potentials + 384 per rule application (2048 flop each).  ks: values of CX_MVC64_K (links per level-0 block); None = the default."""
    model = cx.synth.lgssm_chain(T, d=64, seed=1234)
    out = []
    for k in ks:
        if k is None:
            os.environ.pop("CX_MVC64_K", None)
        else:
            os.environ["CX_MVC64_K"] = str(k)
        dev = cx.DeviceGraph(dim=64, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_into_device(model, dev)
        st = dev.stats()
        dev.sweep(2)
        dt = timed(dev, lambda: dev.sweep(1), steps, 2)
        ps = dev.chain_plan_stats()
        n_mfma = 960 * ps["compositions"] + 384 * ps["rules"]
        tf = n_mfma * 2048 / dt / 1e12
        ref_upd = 5 * T - 4
        parity = check(dev, model) if (check and k is None) else None
        out.append({"config": "C5-scan", "links_per_block": ps["links_per_block"], **({"parity": parity} if parity else {}),
                    "workload": f"d=64 linear-Gaussian chain T={T} ({st['n_edges']} edges), chain-scan schedule: exact forward/backward in one sweep",
                    "ms_per_sweep": dt * 1e3, "reference_updates_per_sweep": ref_upd, "updates_per_s": ref_upd / dt, "plan": ps,
                    "mfma_TFLOPs": tf,
                    "roofline": roofline("mfma", tf, F64_MATRIX_PEAK_TF, "TFLOP/s", None, kernel="k_compose64p + k_walk64b (all launches of one sweep)",
                                         mfma_per_sweep=n_mfma, mfma_per_composition=960, mfma_per_rule=384, compositions=ps["compositions"],
                                         rules=ps["rules"], launches_per_sweep=ps["launches"],
                                         basis="matrix instructions of the plan x 2048 flop / sweep time (launch gaps and the serial top of the tree included)",
                                         minimum_note="the sequential smoother needs 2(T-1) rules = 384 x 2(T-1) matrix instructions; the tree costs "
                                                      f"{n_mfma / (384 * 2 * (T - 1)):.2f}x that for its parallelism")})
        dev.close()
    os.environ.pop("CX_MVC64_K", None)
    return out


def vmp(n=1_000_000, only=None):
    """one variational iteration (all latent states, then both precisions) of the reference's SSM with
    unknown noise precisions, n states: 2n - 1 three-way factors, 6n - 3 edges."""
    model = cx.synth.vmp_ssm(n, seed=1234)
    out = []
    for name, fam in (("structured", L.FAMILY_VMP_STRUCTURED), ("mean_field", L.FAMILY_VMP_MEAN_FIELD)):
        if only and name != only:
            continue
        dev = cx.DeviceGraph(family=fam, schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_vmp_into_device(model, dev)

        def it():
            dev.update_marginals(L.VMP_ALL_NORMAL)
            dev.update_marginals(L.VMP_ALL_PRECISION)
        dt = timed(dev, it, 50, 10)
        g = dev.get_marginals([model.ssnoise, model.obsnoise])
        st = dev.stats()
        out.append({"config": "VMP", "workload": f"{name} VMP, SSM with unknown precisions, n={n} states ({st['n_edges']} edges, {st['n_factors']} factors)",
                    "ms_per_iteration": dt * 1e3, "messages_per_iteration": st["n_messages_per_sweep"],
                    "messages_per_s": st["n_messages_per_sweep"] / dt, "E_ssnoise": g[0, 0] * g[0, 1], "E_obsnoise": g[1, 0] * g[1, 1]})
        tr = vmp_traffic(name) if n == 1_000_000 else None
        if tr:
            out[-1]["roofline"] = roofline("hbm", tr[0] / dt / 1e9, HBM_PEAK_GBS, "GB/s", tr[0],
                                           kernel="all kernels of one iteration (state pass + both precisions)",
                                           basis="counter traffic of one iteration / iteration time", traffic_source=tr[1])
        else:
            out[-1]["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                   "frac_note": "no counter traffic on file for this version of the kernels (tools/profile_vmp.sh)"}
    return out


def vmp_wired(n=100_000, iterations=3, check=None):
    """The structured variational model of the reference's tests as a WIRING (cx_graph_wire under CX_SCHED_REFERENCE: the resolver's
    add_dependency! calls as triples, the rules chosen by the dependency lists) next to the fused family handle that hard-wires the same
    model: per-call times of the replayed plans, what the host pays once, and the agreement of every marginal after `iterations` by-class
    iterations.  (Call-by-call identity with the restated engine, mixed requests included: tests/test_gpu_wired_vmp.py.)"""
    import time
    model = cx.synth.vmp_ssm(n, seed=1234)
    nf = len(model.factor_ids)
    t0 = time.perf_counter()
    t = cx.wiring.structured(model.edge_var, model.edge_fac, model.edge_role, clustered_factors=model.factor_ids[model.n:])
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    dev.graph_create(model.edge_var, model.edge_fac, model.factor_ids, np.full(nf, L.FACTOR_NORMAL_PRECISION, dtype=np.int32), np.zeros(nf), edge_role=model.edge_role)
    dev.graph_wire(t.signals, t.dependencies, t.flags)
    wire_s = time.perf_counter() - t0
    dev.set_marginals([model.ssnoise, model.obsnoise], L.FORM_GAMMA, [1.0, 1.0, 1.0, 1.0])
    dev.set_marginals(model.x_ids, L.FORM_MEAN_PRECISION, np.tile([0.0, 1.0], model.n))
    dev.set_marginals(model.y_ids, L.FORM_POINT, model.data_y)
    fused = cx.DeviceGraph(family=L.FAMILY_VMP_STRUCTURED, schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_vmp_into_device(model, fused)
    calls = (("states", model.x_ids, L.VMP_ALL_NORMAL), ("ssnoise", [model.ssnoise], [model.ssnoise]), ("obsnoise", [model.obsnoise], [model.obsnoise]))
    t0 = time.perf_counter()
    for _ in range(iterations):
        for _name, ids, fids in calls:
            dev.sweep_for(ids); fused.update_marginals(fids)
    dev.sync()
    first_s = time.perf_counter() - t0
    xa, xb = dev.get_marginals(model.x_ids), fused.get_marginals(model.x_ids)
    ga, gb = dev.get_marginals([model.ssnoise, model.obsnoise]), fused.get_marginals([model.ssnoise, model.obsnoise])
    err = max(float(np.max(np.abs(xa[:, 0] - xb[:, 0]) / np.maximum(np.abs(xb[:, 0]), 1e-3))), float(np.max(np.abs(1.0 / xa[:, 1] - xb[:, 1]) / xb[:, 1])),
              float(np.max(np.abs(ga - gb) / np.abs(gb))))
    per_call, plans = {}, {}
    for i, (name, ids, _f) in enumerate(calls):
        best = None
        for _rep in range(3):
            for _n2, ids2, _f2 in calls[:i]:
                dev.sweep_for(ids2)
            dev.sync(); t0 = time.perf_counter(); dev.sweep_for(ids); dev.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            st = dev.ref_plan_stats()
            for _n2, ids2, _f2 in calls[i + 1:]:
                dev.sweep_for(ids2)
        per_call[name] = best * 1e3
        plans[name] = {k: st[k] for k in ("stages", "launches", "executions", "messages", "rounds")}
    st = dev.ref_plan_stats()
    dtf = timed(fused, lambda: [fused.update_marginals(f) for _n, _i, f in calls], 20, 5)
    execs = sum(p["executions"] for p in plans.values())
    total_ms = sum(per_call.values())
    return {"config": "VMP-wired", "workload": f"structured VMP as a user wiring (cx_graph_wire, {len(t.flags)} add_dependency! triples), SSM with unknown precisions, n={n} states",
            "ms_per_call": per_call, "ms_per_iteration": total_ms, "plans": plans, "executions_per_iteration": execs, "executions_per_s": execs / (total_ms * 1e-3),
            "host_once": {"triples_and_wiring_s": wire_s, "first_iterations_s": first_s, "plans_kept": st["plans"], "calls_that_ran_the_scheduler": st["misses"]},
            "fused_family_ms_per_iteration": dtf * 1e3,
            "roofline": {"bound": "hbm", "achieved": execs * 32 / (total_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                         "frac_survey_convention": execs * 32 / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "basis": "algorithmic bytes (32 B per execution, SURVEY §8d) / time of the three calls",
                         "frac_note": "the states' call is the reference's forward / backward chain: 2 n dependent stages inside one workgroup, a latency chain by construction; "
                                      "the precisions' calls are wide and shallow", "kernel": "k_batch_run / k_batch over the plan's stages"},
            # device against device: NOT parity (bench.py's hook adds the `parity` object: the CPU checker's array form at this size)
            "self_check": {"max_rel_err": err, "tolerance": 1e-9, "ok": bool(err <= 1e-9), "checker": "the fused family handle on the same device (cx_update_marginals), every state mean, "
                           "state precision and both Gamma marginals", "sample": f"after {iterations} by-class iterations, n={n}"},
            **({"parity": check(xa, ga, model, iterations)} if check else {})}


def vmp_traffic(family):
    """HBM bytes of one iteration of a variational family from the newest profiles/r*_vmp_rocprof.json (tools/profile_vmp.sh) whose
    kernel sources are the current ones; None otherwise"""
    import glob
    from importlib import import_module
    sha = import_module("cortex.jl_amd.build").sources_sha16
    want = sha("k_mf_normal") + sha("k_chain_apply")
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_vmp_rocprof.json"))):
        try:
            d = json.load(open(f)).get(family, {})
        except Exception:
            continue
        if d.get("sources_sha16") == want and d.get("hbm_bytes_per_iteration"):
            best = (float(d["hbm_bytes_per_iteration"]), os.path.relpath(f, ROOT))
    return best


def _tree_traffic(row, n_factors):
    """counter bytes of one exact tree sweep from profiles/r*_tree_traffic.json (tools/profile_tree.sh), when they were taken on this
    version of the item and scan kernels and on the same forest"""
    import glob
    from importlib import import_module
    sha = import_module("cortex.jl_amd.build").sources_sha16
    want = sha("k_batch") + sha("k_chain_")
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tree_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        r = d.get("rows", {}).get(row, {})
        if r.get("sources_sha16") == want and r.get("hbm_bytes_per_sweep") and d.get("n_factors", {}).get(row) == n_factors:
            best = (float(r["hbm_bytes_per_sweep"]), os.path.relpath(f, ROOT))
    return best


def tree(n_factors=200_000, steps=20, shape="random", check=None):
    """the tree schedule (CX_SCHED_TREE): ONE sweep on a forest = the reference's one update_marginals! there, level by level.  Not a
    BASELINE config: a bushy tree with factors of 2..6 variables and ~10^6 edges, timed per exact sweep, beside the number of fused
    sweeps the fixed-point schedule needs for the same result on the same graph."""
    model = cx.synth.tree_model(n_factors, seed=31, shape=shape, observe=0.2)
    dev = cx.DeviceGraph(schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(2)
    dt = timed(dev, lambda: dev.sweep(1), steps, 3)
    st = dev.tree_plan_stats()
    hp = dev.tree_heavy_path_stats()       # zeros when the sweep runs level by level
    launches = hp["launches"] or st["stages"]
    fused = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, fused)
    need = 2 * st["depth"] + 2
    fused.sweep(need)
    dtf = timed(fused, lambda: fused.sweep(1), steps, 3)
    ids = model.x_ids[:: max(len(model.x_ids) // 100_000, 1)]
    a, b = dev.get_marginals(ids), fused.get_marginals(ids)
    err = float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), np.median(np.abs(b)))))
    n_msgs = st["messages_up"] + st["messages_down"]
    tt = _tree_traffic("tree" if shape == "random" else f"tree-{shape}", n_factors)
    return {"config": "tree" if shape == "random" else f"tree-{shape}", "workload": f"scalar Gaussian forest, {n_factors} factors of 2..6 variables + a prior per variable ({len(model.edge_var)} edges, shape {shape})",
            "ms_per_sweep": dt * 1e3, "plan": st, "heavy_paths": hp, "launches_per_sweep": launches, "messages_per_sweep": n_msgs, "messages_per_s": n_msgs / dt,
            "fused_schedule": {"ms_per_sweep": dtf * 1e3, "sweeps_to_the_same_result": need, "ms_to_the_same_result": need * dtf * 1e3},
            "roofline": roofline("hbm", (tt[0] if tt else n_msgs * 32) / dt / 1e9, HBM_PEAK_GBS, "GB/s", tt[0] if tt else None,
                                 kernel="k_chain_* scans of the heavy paths + k_batch item stages" if hp["launches"] else "k_batch, one launch per stage",
                                 basis=(f"counter bytes per sweep (FETCH_SIZE x2 + WRITE_SIZE over all launches of a sweep, {tt[1]}) / sweep time" if tt
                                        else "algorithmic bytes (32 B per message, SURVEY §8d) / sweep time"),
                                 frac_survey_convention=n_msgs * 32 / dt / 1e9 / HBM_PEAK_GBS,
                                 frac_note="a latency-bound schedule" + ("" if tt else ": the fraction of HBM on SURVEY §8d's byte convention is frac_survey_convention, no counter traffic on file for these sources and this forest"),
                                 bound_detail=f"not a bandwidth-bound schedule: {launches} dependent launches (≈ {dt / max(launches, 1) * 1e6:.1f} us "
                                              "each at this size): the time is the number of dependent launches x per-kernel time"),
            "self_check": {"max_rel_err_marginals": err, "ok": bool(err < 1e-9), "checker": "the fused schedule at its fixed point on the same device",
                           "sample": f"{len(ids)} marginals"},
            **({"parity": check(dev, model)} if check else {})}


def tree_mv(d=4, n_spine=200_000):
    """the tree schedule for dim 2..4 over heavy paths: a d-dimensional chain of n_spine states with a latent state below each (depth
    ~ n_spine levels), ONE exact sweep; beside it the level schedule on a tenth of the model (its time is launches x per-launch time)"""
    import os
    model = cx.synth.lgssm_comb(n_spine, d=d, teeth=1, seed=5)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(model, dev)
    dev.sweep(2)
    dt = timed(dev, lambda: dev.sweep(1), 20, 3)
    st, hp = dev.tree_plan_stats(), dev.tree_heavy_path_stats()
    n_msgs = st["messages_up"] + st["messages_down"]
    small = cx.synth.lgssm_comb(n_spine // 10, d=d, teeth=1, seed=5)
    os.environ["CX_TREE_HP"] = "0"
    lv = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(small, lv)
    lv.sweep(1)
    dtl = timed(lv, lambda: lv.sweep(1), 3, 1)
    os.environ["CX_TREE_HP"] = "1"
    hs = cx.DeviceGraph(dim=d, schedule=L.SCHED_TREE)
    cx.synth.load_into_device(small, hs)
    hs.sweep(1)
    del os.environ["CX_TREE_HP"]
    ids = small.x_ids[:: max(len(small.x_ids) // 50_000, 1)]
    a, b = hs.get_marginals(ids), lv.get_marginals(ids)
    err = float(np.nanmax(np.abs(a - b)) / np.nanmax(np.abs(b)))
    S = (d + d * d) * 8
    return {"config": "tree-mv", "workload": f"d={d} linear-Gaussian chain of {n_spine} states with a latent state below each ({len(model.edge_var)} edges), tree schedule over heavy paths",
            "ms_per_sweep": dt * 1e3, "plan": st, "heavy_paths": hp, "launches_per_sweep": hp["launches"] or st["stages"], "messages_per_sweep": n_msgs, "messages_per_s": n_msgs / dt,
            "level_schedule_on_a_tenth": {"ms_per_sweep": dtl * 1e3, "stages": lv.tree_plan_stats()["stages"]},
            "roofline": roofline("hbm", n_msgs * 2 * S / dt / 1e9, HBM_PEAK_GBS, "GB/s", None, kernel="k_mvc_* scans of the heavy paths + k_batch_mv item stages" if d <= 4 else "k_compose64p / k_walk64b plans of the heavy paths + k_rule64w item stages",
                                 basis=f"algorithmic bytes ({2 * S} B per message, SURVEY §8d) / sweep time", frac_survey_convention=n_msgs * 2 * S / dt / 1e9 / HBM_PEAK_GBS,
                                 frac_note="a latency-bound schedule: the fraction of HBM on SURVEY §8d's byte convention is frac_survey_convention, no counter traffic was collected"),
            # device against device (the joint solves of this schedule: tests/test_gpu_tree.py); not a parity claim
            "self_check": {"max_rel_err_marginals": err, "ok": bool(err < 1e-8), "checker": "the level schedule on the same device, a tenth of the model",
                           "sample": f"{len(ids)} marginals"}}


def reference_order(n=1415, calls_timed=5, tol=1e-9, max_calls=400, check=None, fixed_point=True):
    """C4 under CX_SCHED_REFERENCE: ONE cx_sweep = ONE update_marginals! of the reference on the loopy grid (the same executions in the
    same order, csrc/cx_refsched.h).  Reported: the plan (stages, launches, executions), the host's one-off cost of finding it, ms per
    call when a standing plan is replayed, and how many calls / fused sweeps it takes until no message moves by more than `tol` —
    the sequential pass reads new values and needs fewer iterations, each of which costs thousands of dependent stages."""
    model = cx.synth.gaussian_grid(n, n, seed=1234)
    prior = np.stack([model.prior_mean, model.prior_variance], axis=1)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    t0 = time.perf_counter()
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    t_load = time.perf_counter() - t0
    plan_s = []
    for _ in range(2):          # the first call starts from the seeded state, the second from the state every later call starts from
        t0 = time.perf_counter()
        dev.sweep(1); dev.sync()
        plan_s.append(time.perf_counter() - t0)
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
    t_call, t_set = [], []
    for _ in range(calls_timed):
        dev.sync()
        t0 = time.perf_counter()
        dev.sweep(1); dev.sync()
        t_call.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
        t_set.append(time.perf_counter() - t0)
    st = dev.ref_plan_stats()
    parity = check(dev, model) if check else None
    out = {"config": "C4-reference", "workload": f"{n}x{n} Gaussian grid ({len(model.edge_var)} edges), CX_SCHED_REFERENCE: one cx_sweep = one update_marginals! of the reference (sequential order, newest values)",
           "ms_per_call": float(np.median(t_call)) * 1e3, "plan": st, "executions_per_call": st["executions"], "messages_per_s": st["messages"] / float(np.median(t_call)),
           "host": {"graph_upload_and_wiring_s": t_load, "first_two_calls_s_scheduler_levelling_upload_capture": plan_s,
                    "re_setting_the_priors_s_per_call": float(np.median(t_set))},
           "roofline": roofline("hbm", st["messages"] * 32 / float(np.median(t_call)) / 1e9, HBM_PEAK_GBS, "GB/s", None,
                                kernel="k_ref_cluster: the plan's stages behind single-XCD barriers in one launch (stages wider than 65,536 items: k_batch on the whole chip); "
                                       "CX_REF_CLUSTER=0: k_batch / k_batch_run, one launch per stage or run of thin stages",
                                basis="algorithmic bytes (32 B per message, SURVEY §8d) / call time", frac_survey_convention=st["messages"] * 32 / float(np.median(t_call)) / 1e9 / HBM_PEAK_GBS,
                                frac_note="a latency-bound schedule by construction: the reference's order is sequential, its dependency depth is the stage count",
                                bound_detail=f"{st['stages']} dependent stages in {st['launches']} launches (≈ {float(np.median(t_call)) / max(st['stages'], 1) * 1e6:.2f} us per stage)")}
    if parity:
        out["parity"] = parity
    if fixed_point:
        # calls until no message moves by more than tol (cx_residual between consecutive calls), beside the fused schedule's sweeps
        fresh = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        cx.synth.load_into_device(model, fresh, seed_variance=1e6)
        fresh.residual()
        calls, res = 0, float("inf")
        while calls < max_calls and not res <= tol:
            if calls:
                fresh.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
            fresh.sweep(1)
            calls += 1
            res = fresh.residual()
        fused = cx.DeviceGraph(schedule=L.SCHED_FUSED)
        cx.synth.load_into_device(model, fused, seed_variance=1e6)
        fused.residual()
        sweeps, resf = 0, float("inf")
        while sweeps < 4 * max_calls and not resf <= tol:
            fused.sweep(1)
            sweeps += 1
            resf = fused.residual()
        dtf = timed(fused, lambda: fused.sweep(1), 20, 2)
        ids = model.x_ids[:: max(len(model.x_ids) // 200_000, 1)]
        a, b = fresh.get_marginals(ids), fused.get_marginals(ids)
        out["to_the_fixed_point"] = {"tolerance_max_abs_change_of_a_message": tol, "reference_order_calls": calls, "fused_sweeps": sweeps,
                                     "reference_order_ms": calls * out["ms_per_call"], "fused_ms": sweeps * dtf * 1e3, "fused_ms_per_sweep": dtf * 1e3,
                                     "marginals_agree_to": float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), np.median(np.abs(b)))))}
        fresh.close(); fused.close()
    dev.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3", "c5"]
    torch.cuda.init()
    for w in which:
        if w.startswith("reference"):                    # reference | reference:300 (grid side)
            parts = w.split(":")                           # reference:1415:nofp skips the iteration to the fixed point
            print(json.dumps(reference_order(n=int(parts[1]) if len(parts) > 1 else 1415, fixed_point="nofp" not in parts)), flush=True)
            continue
        if w.startswith("vmpwired"):                     # vmpwired | vmpwired:1000000
            print(json.dumps(vmp_wired(int(w.split(":")[1]) if ":" in w else 100_000)), flush=True)
            continue
        if w.startswith("vmp"):
            for r in vmp(only=w[4:] or None):          # vmp | vmp_structured | vmp_mean_field
                print(json.dumps(r), flush=True)
            continue
        if w.startswith("treemv"):                       # treemv | treemv:2 | treemv:64 (5,000 states on the spine)
            dd = int(w.split(":")[1]) if ":" in w else 4
            print(json.dumps(tree_mv(d=dd, n_spine=200_000 if dd <= 4 else 5_000)), flush=True)
            continue
        if w.startswith("tree"):                         # tree | tree:deep
            print(json.dumps(tree(shape=w.split(":")[1] if ":" in w else "random")), flush=True)
            continue
        if w.startswith("tilesn"):                       # tilesn:16 — the native kernel's sweeps alone (the profilers' form)
            print(json.dumps(mv_tiles(int(w.split(":")[1]) if ":" in w else 16, native_only=True)), flush=True)
            continue
        if w.startswith("tiles"):                        # tiles | tiles:8 | tiles:32 (the user's dim)
            print(json.dumps(mv_tiles(int(w.split(":")[1]) if ":" in w else 16)), flush=True)
            continue
        if w.startswith("c5scan"):                       # c5scan | c5scan:49,98,196 (links per level-0 block)
            ks = tuple(int(x) for x in w.split(":")[1].split(",")) if ":" in w else (None,)
            for r in mv64_scan(100_000, 10, ks):
                print(json.dumps(r), flush=True)
            continue
        if w.startswith("c3scan"):                       # c3scan | c3scan:1,2,4,8 (links per thread)
            ks = tuple(int(x) for x in w.split(":")[1].split(",")) if ":" in w else (None,)
            for r in mv_scan(4, 1_000_000, 30, ks):
                print(json.dumps(r), flush=True)
            continue
        r = c2() if w == "c2" else (mv(4, 1_000_000, 30) if w == "c3" else mv(64, 100_000, 20))
        print(json.dumps(r), flush=True)
