import os, sys
sys.path.insert(0, os.getcwd())
import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
model = cx.synth.lgssm_chain(100_000, d=64, seed=1234)
dev = cx.DeviceGraph(dim=64, schedule=L.SCHED_CHAIN_SCAN)
cx.synth.load_into_device(model, dev)
dev.sweep(1); dev.sync()
print("---- second sweep", file=sys.stderr)
dev.sweep(1); dev.sync()
