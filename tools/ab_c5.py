#!/usr/bin/env python3
"""tools/ab_c5.py — the d = 64 rule kernel forms side by side in ONE process on ONE device (interleaved rounds; devices differ by
more than 10 % on matrix-dense kernels, so numbers from different boxes must not be ranked): workgroup per message (shipped
round 1) vs wave per message (CX_RULE64=w).  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
model = cx.synth.lgssm_chain(T, d=64, seed=1234)
dev = cx.DeviceGraph(dim=64, schedule=L.SCHED_FUSED)
cx.synth.load_into_device(model, dev, seed_variance=1e6)
dev.sweep(2)
forms = ["wg", "wave"]
res = {k: [] for k in forms}
for rnd in range(4):
    for form in forms:
        os.environ["CX_RULE64"] = "g" if form == "wg" else "w"
        dev.sweep(2); dev.sync()
        t0 = time.perf_counter()
        dev.sweep(6); dev.sync()
        res[form].append((time.perf_counter() - t0) / 6 * 1e3)
out = {k: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v)), "all": [round(x, 3) for x in v]} for k, v in res.items()}
out = {k: round(v["median_ms"], 3) for k, v in out.items()}
out["wave_over_wg"] = out["wave"] / out["wg"]
print(json.dumps(out))
