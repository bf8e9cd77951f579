"""lab: two d = 16 handles one after the other, 60 sweeps each, under rocprofv3 --kernel-trace: is the ~70 ms stall of the second handle
(tools/lab/tiles_after_c5.py) one long kernel or a gap between kernels?
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/stall -- python3 $R/tools/lab/stall_trace.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

for h in range(2):
    model = cx.synth.lgssm_chain(100_000, d=16, seed=1234)
    dev = cx.DeviceGraph(dim=16, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(2)
    dev.sync()
    for b in range(3):
        t0 = time.perf_counter()
        dev.sweep(20)
        dev.sync()
        print(f"handle {h} batch {b}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per sweep", flush=True)
    dev.close()
