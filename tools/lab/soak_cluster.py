"""Soak of the XCD-resident cluster (cx_batch.hip: k_ref_cluster) at C4 size: the same handle three times — twice on the cluster, once as
plain launches (CX_REF_CLUSTER=0) — through N consecutive "set the priors, call" iterations of the reference-order schedule.  The two
cluster handles run the same code on the same inputs, so every stored message and marginal has to agree BIT FOR BIT after every call,
whichever XCD and workgroups each launch landed on: a value read before its writer's store had reached the L2, or a barrier let go
early, would show as a difference.  Against the launches (other kernels: the compiler contracts their multiply-adds differently) the
difference has to stay at rounding level."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

side, calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1415, int(sys.argv[2]) if len(sys.argv) > 2 else 150
model = cx.synth.gaussian_grid(side, side, seed=1)
prior = np.stack([model.prior_mean, model.prior_variance], axis=1)


def make():
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    return dev


a = make()
a2 = make()
os.environ["CX_REF_CLUSTER"] = "0"
b = make()
del os.environ["CX_REF_CLUSTER"]
rng = np.random.default_rng(0)
sample = rng.choice(len(model.edge_var), size=400000, replace=False)
worst, differing_calls, t_a, t_b, t_a2 = 0.0, 0, [], [], []
for call in range(calls):
    for dev, ts in ((a, t_a), (a2, t_a2), (b, t_b)):
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
        dev.sync()
        t0 = time.perf_counter()
        dev.sweep(1)
        dev.sync()
        ts.append(time.perf_counter() - t0)
    ma, mb = a.get_marginals(model.x_ids), b.get_marginals(model.x_ids)
    fa = a.get_messages(model.edge_var[sample], model.edge_fac[sample], L.TO_VARIABLE, L.FORM_NATURAL)
    fb = b.get_messages(model.edge_var[sample], model.edge_fac[sample], L.TO_VARIABLE, L.FORM_NATURAL)
    ma2 = a2.get_marginals(model.x_ids)
    fa2 = a2.get_messages(model.edge_var[sample], model.edge_fac[sample], L.TO_VARIABLE, L.FORM_NATURAL)
    if not (np.array_equal(ma, ma2, equal_nan=True) and np.array_equal(fa, fa2, equal_nan=True)):
        differing_calls += 1
    scale = max(1.0, float(np.nanmax(np.abs(mb))))
    worst = max(worst, float(np.nanmax(np.abs(ma - mb))) / scale, float(np.nanmax(np.abs(fa - fb)) / max(1.0, float(np.nanmax(np.abs(fb))))))
    if call % 25 == 24:
        print(f"# call {call + 1}: differing calls so far {differing_calls}", file=sys.stderr, flush=True)
st = a.cluster_stats()
print(json.dumps({"grid": f"{side}x{side}", "calls": calls, "compared_per_call": f"all {len(model.x_ids)} marginals + {len(sample)} sampled factor->variable messages, bit for bit",
                  "calls_where_the_two_cluster_handles_differ_in_any_bit": differing_calls, "cluster_vs_launches_max_difference_relative_to_the_largest_value": worst, "cluster_state_at_the_end": st["state"],
                  "ms_per_call_cluster_median": 1e3 * float(np.median(t_a[2:])), "ms_per_call_cluster_max": 1e3 * float(np.max(t_a[2:])),
                  "ms_per_call_launches_median": 1e3 * float(np.median(t_b[2:]))}))
