"""lab: one exact sweep (chain-scan schedule) of a d-dimensional chain of T states in its native tile size and embedded in 64
python tools/lab/nt_chain_time.py [d] [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
for name, env, Tn in (("native", None, T), ("embedded in 64", "64", min(T, 20_000))):
    model = cx.synth.lgssm_chain(Tn, d=d, seed=1234)
    if env:
        os.environ["CX_MFMA_DIM"] = env
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_CHAIN_SCAN)
    os.environ.pop("CX_MFMA_DIM", None)
    cx.synth.load_into_device(model, dev)
    dev.sweep(2)
    dev.sync()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        dev.sweep(5)
        dev.sync()
        out.append((time.perf_counter() - t0) / 5 * 1e3)
    st = dev.chain_plan_stats()
    print(f"d = {d}, {name}, T = {Tn}: {sorted(out)[1]:.3f} ms per exact sweep ({sorted(out)[1] * 1e6 / Tn:.1f} ns per state); plan {st}; device bytes {dev.stats()['device_bytes']}", flush=True)
    dev.close()
