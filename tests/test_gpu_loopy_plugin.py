"""-m gpu: the plug-in on LOOPY graphs, call by call against the restated reference engine (oracle/cortex_ref.c).

On a graph with cycles ONE update_marginals! of the reference is a sequential pass that always reads the newest values
(src/inference_engine.jl:575-608, src/signal.jl:466-490: requested-variable order x neighbour order).  What it leaves after a call is
NOT what a flooding / fused device sweep leaves — only the fixed point is shared (tests/test_oracle_reference_kats.py:735) — so the
per-call comparison goes through the modes that keep the reference's order: "per_signal" (the host scheduler drives one launch per
signal) and "wavefront", which finds the cycle in the request's dependency graph and hands the call back to the host scheduler.
(tests/test_loopy_reference_order.py runs the same comparison on the CPU with a host processor: the harness itself is checked there.)"""
import pytest

from tests.loopy_support import HipBackend, models, run_calls

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["grid8x9", "grid48x40", "random"])
def test_per_signal_mode_on_a_loopy_graph_is_the_reference_call_by_call(hip_lib, name):
    b = HipBackend("per_signal")
    executed = run_calls(models()[name], b)
    assert min(executed) > 0 and (name == "random" or executed[1] == executed[2])      # (hubs of degree > 5: the readiness flags settle later)


@pytest.mark.parametrize("name", ["grid8x9", "random"])
def test_wavefront_mode_on_a_loopy_graph_hands_the_call_back_to_the_host_scheduler(hip_lib, name):
    """a wavefront of "all pending signals" on a loopy graph would be a Jacobi step (every signal reads OLD values); the reference reads
    NEW ones.  run_wavefronts is only ever entered for requests whose dependency graph is acyclic: HipProcessor finds the cycle and lets
    the generic scheduler run the call in the reference's order — same results, same order, one launch per signal."""
    b = HipBackend("wavefront")
    run_calls(models()[name], b)
    assert b.proc.cyclic_requests == 3


def test_a_partial_request_in_reverse_order(hip_lib):
    """a request that names some variables, in descending id order: the lazy scheduler computes what those marginals need, in its order"""
    model = models()["grid8x9"]
    run_calls(model, HipBackend("per_signal"), request=model.x_ids[::-3].copy())
