"""lab: when does the ~70 ms gap (tools/lab/stall_trace.py: a gap between two kernels of one queue) appear?  (a) second handle while the
first stays open and nothing has been freed in the process; (b) after the first was closed."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402


def make():
    model = cx.synth.lgssm_chain(100_000, d=16, seed=1234)
    dev = cx.DeviceGraph(dim=16, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(2)
    dev.sync()
    return dev


def run(tag, dev, batches=8):
    out = []
    for _ in range(batches):
        t0 = time.perf_counter()
        dev.sweep(20)
        dev.sync()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"{tag}: " + " ".join(f"{x:.2f}" for x in out), flush=True)


a = make()
run("first handle", a)
b = make()
run("second handle, the first still open, nothing freed yet", b)
run("the first handle again", a)
c = make()
run("third handle, both others open", c)
a.close()
run("second handle after the first was closed", b)
d = make()
run("a new handle after a close", d)
