"""ms per update_marginals! of the reference-order schedule on a d-dimensional linear-Gaussian state-space chain (the C3 model, shorter):
new data, one cx_sweep — the plan of the steady state replayed.  CX_REF_RUN_MAX=0: every stage a launch (A/B of the runs of thin stages)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

d, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
model = cx.synth.lgssm_chain(T, d=d, seed=1)
dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_REFERENCE)
cx.synth.load_into_device(model, dev)
y = np.asarray(model.data_y)
times = []
for it in range(6):
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y + 0.01 * it)
    dev.sync()
    t0 = time.perf_counter()
    dev.sweep(1)
    dev.sync()
    times.append(time.perf_counter() - t0)
st = dev.ref_plan_stats()
scan = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
cx.synth.load_into_device(model, scan)
scan.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y + 0.05)
scan.sweep(1)
err = float(np.max(np.abs(dev.get_marginals(model.x_ids) - scan.get_marginals(model.x_ids))))
print(json.dumps({"d": d, "T": T, "ms_per_call": 1e3 * float(np.median(times[2:])), "plan": st, "max_abs_vs_chain_scan": err}))
