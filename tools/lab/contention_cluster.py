"""The XCD-resident cluster beside another process's kernels on the same GPU: a child process runs fused sweeps on a C4-sized grid without
pause while this process makes reference-order calls (one cluster launch each).  The cluster's workgroups have to be resident together;
another tenant's short kernels only delay their arrival.  Reported: ms per call alone and under contention, and whether any call failed
(a barrier that times out fails the call loudly: cx_api_ref.hip: cluster_run)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "tenant":
    model = cx.synth.gaussian_grid(1415, 1415, seed=2)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    t_end = time.time() + float(sys.argv[2])
    n = 0
    while time.time() < t_end:
        dev.sweep(200); dev.sync(); n += 200
    print(json.dumps({"tenant_sweeps": n}), flush=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "tenant_ref":      # a second user of the cluster: reference-order calls of its own
    model = cx.synth.gaussian_grid(1000, 1000, seed=2)
    prior = np.stack([model.prior_mean, model.prior_variance], axis=1)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    t_end = time.time() + float(sys.argv[2])
    n = failed = 0
    while time.time() < t_end:
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
        try:
            dev.sweep(1); dev.sync(); n += 1
        except cx.CortexHipError:
            failed += 1
    print(json.dumps({"tenant_reference_calls": n, "tenant_failed": failed, "tenant_cluster_state": dev.cluster_stats()["state"]}), flush=True)
    sys.exit(0)

side, calls = 1415, 40
model = cx.synth.gaussian_grid(side, side, seed=1)
prior = np.stack([model.prior_mean, model.prior_variance], axis=1)
dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
cx.synth.load_into_device(model, dev, seed_variance=1e6)


def run(n):
    ts, failed = [], 0
    for _ in range(n):
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
        dev.sync()
        t0 = time.perf_counter()
        try:
            dev.sweep(1); dev.sync()
        except cx.CortexHipError as e:
            failed += 1
            print("# failed:", str(e)[:160], file=sys.stderr, flush=True)
        ts.append(time.perf_counter() - t0)
    return ts, failed


run(4)
alone, f0 = run(calls)
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), os.environ.get("TENANT", "tenant"), "25"], stdout=subprocess.PIPE, text=True)
time.sleep(12.0)      # the tenant imports, builds its grid and starts sweeping
shared, f1 = run(calls)
out, _ = child.communicate(timeout=120)
marg = dev.get_marginals(model.x_ids)
print(json.dumps({"grid": f"{side}x{side}", "calls": calls, "ms_per_call_alone_median": 1e3 * float(np.median(alone)), "failed_alone": f0,
                  "ms_per_call_beside_a_tenant_median": 1e3 * float(np.median(shared)), "ms_per_call_beside_a_tenant_max": 1e3 * float(np.max(shared)),
                  "failed_beside_a_tenant": f1, "tenant": json.loads(out.strip().splitlines()[-1]) if out.strip() else None,
                  "cluster_state_at_the_end": dev.cluster_stats()["state"], "marginals_finite": bool(np.all(np.isfinite(marg)))}))
