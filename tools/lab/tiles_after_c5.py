"""lab: a d = 16 handle's sweeps, alone and after a d = 64 handle has lived in the same process (bench.py saw 4.3 ms of wall time per
sweep around a 0.40 ms kernel for the native-tile row behind the C5 row): the time of every batch of 20 sweeps"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402


def native(tag, batches=12):
    model = cx.synth.lgssm_chain(100_000, d=16, seed=1234)
    dev = cx.DeviceGraph(dim=16, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(2)
    dev.sync()
    out = []
    for _ in range(batches):
        t0 = time.perf_counter()
        dev.sweep(20)
        t1 = time.perf_counter()
        dev.sync()
        out.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
    print(tag, "host ms to enqueue 20 | ms per sweep:", " ".join(f"{a:.2f}|{b:.2f}" for a, b in out), flush=True)
    dev.close()


def big(T, sweeps):
    model = cx.synth.lgssm_chain(T, d=64, seed=1234)
    dev = cx.DeviceGraph(dim=64, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(sweeps)
    dev.sync()
    dev.close()
    print(f"(a d = 64 handle of T = {T} swept {sweeps} times and closed)", flush=True)


native("alone")
big(2_000, 3)
native("after d = 64 (closed)")
big(2_000, 3)
time.sleep(1.0)
native("after d = 64 (closed) and a second's sleep")
import gc
big(2_000, 3)
gc.collect()
native("after d = 64 (closed) and gc.collect()")
keep = cx.DeviceGraph(dim=64, schedule=L.SCHED_FUSED)
cx.synth.load_into_device(cx.synth.lgssm_chain(2_000, d=64, seed=1234), keep, seed_variance=1e6)
keep.sweep(3)
keep.sync()
native("beside a d = 64 handle that stays open")
native("again, beside the open handle (the last d = 16 handle was closed)")
