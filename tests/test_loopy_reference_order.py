"""CPU: the host mirror's scheduler (cortex.jl_amd/{signal,inference_engine,dependencies}.py — the part of the reference the host
keeps) against the C restatement of the reference (oracle/cortex_ref.c) on LOOPY graphs, call by call: identical execution order
and identical values over consecutive update_marginals! calls with the priors re-set in between.  The GPU twin of this file
(tests/test_gpu_loopy_plugin.py) swaps the host processor for HipProcessor; the comparison code is shared (tests/loopy_support.py)."""
import numpy as np
import pytest

from cortex.jl_amd.hip_processor import request_has_cycle
from cortex.jl_amd.inference_engine import request_inference_for
from tests.helpers import mirror_engine_from_model
from tests.loopy_support import HostBackend, models, run_calls


@pytest.mark.parametrize("name", ["grid8x9", "grid48x40", "random"])
def test_host_mirror_and_c_restatement_agree_call_by_call_on_loopy_graphs(name):
    model = models()[name]
    executed = run_calls(model, HostBackend(), n_calls=3 if name.startswith("grid") else 6, rtol=1e-12)
    n_pairwise_directed = 2 * int(np.sum(model.factor_kind == 1)) * 2      # both directions, both message kinds
    deg = np.bincount(model.edge_var)
    n_products = executed[1] - n_pairwise_directed - len(model.x_ids)      # segment-tree intermediates of variables of degree > 5
    if deg.max() <= 5:
        assert executed[1] == executed[2] and n_products == 0, "on the grids a call after the first computes every listened message once"
    else:       # the readiness flags of a graph with hubs settle later: the number of executions still moves from call to call
        assert n_products > 0 and min(executed) > 0, executed


def test_partial_request_in_reverse_order_on_a_loopy_graph():
    model = models()["grid8x9"]
    run_calls(model, HostBackend(), request=model.x_ids[::-3].copy(), rtol=1e-12)


def test_cycle_detection_of_the_wavefront_mode():
    import cortex.jl_amd as cx
    from tests.test_host_mirror import SSMBeliefPropagationProcessor, make_ssm

    for name, model in models().items():
        e = mirror_engine_from_model(model, cx.InferenceRequestScanner())
        assert request_has_cycle(request_inference_for(e, [int(v) for v in model.x_ids])), name
    e, x, *_ = make_ssm(50, SSMBeliefPropagationProcessor())
    assert not request_has_cycle(request_inference_for(e, x))
    tree = cx.synth.tree_model(40, seed=3, observe=0.0)
    e = mirror_engine_from_model(tree, cx.InferenceRequestScanner())
    assert not request_has_cycle(request_inference_for(e, [int(v) for v in tree.x_ids]))
