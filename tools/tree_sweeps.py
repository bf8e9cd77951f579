"""N exact sweeps of the tree schedule on the forest of bench.py's `tree` rows (tools/profile_tree.sh runs it under rocprofv3 with two
values of N: the difference of the counters is N2 - N1 sweeps' traffic, whatever the load and the first sweeps cost).
python3 tools/tree_sweeps.py random|deep N_FACTORS N_SWEEPS"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cortex.jl_amd as cx  # noqa: E402
from cortex.jl_amd import _lib as L  # noqa: E402

shape, n_factors, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
model = cx.synth.tree_model(n_factors, seed=31, shape=shape, observe=0.2)
dev = cx.DeviceGraph(schedule=L.SCHED_TREE)
cx.synth.load_into_device(model, dev)
dev.sweep(2)
dev.sync()
dev.sweep(n)
dev.sync()
st = dev.tree_plan_stats()
print(f"{shape} {n_factors} factors, {len(model.edge_var)} edges: {n} sweeps after 2; messages per sweep {st['messages_up'] + st['messages_down']}")
