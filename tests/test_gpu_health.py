"""-m gpu: cx_message_health — the numerical guards of the boundary as counters (non-finite values, variances <= 0 and
matrices that are not positive definite are reported through status codes and counters, never by an abort).  Counted on the device:
the stored factor→variable messages into non-observed variables that are defined / undefined (UndefValue = NaN) / defined with a
negative precision / defined with a non-finite entry."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L

pytestmark = pytest.mark.gpu


def _readable(model):
    obs = set(int(v) for v in model.data_var)
    return int(sum(int(v) not in obs for v in model.edge_var))


def test_scalar_messages_are_undefined_until_a_sweep_computes_them(hip_lib):
    m = cx.synth.ssm_chain(500, seed=3)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(m, dev)
    n = _readable(m)
    before = dev.message_health()
    assert before["defined"] + before["undefined"] == n and before["undefined"] > 0      # nothing computed yet (the reference: UndefValue())
    dev.sweep(1)
    after = dev.message_health()
    assert after == {"defined": n, "undefined": 0, "negative_precision": 0, "non_finite": 0}
    # a message a user broke: a negative precision and an infinite mean are counted, not propagated silently as "fine"
    x, lik = int(m.x_ids[7]), None
    ev, ef = m.edge_var, m.edge_fac
    facs = [int(f) for v, f in zip(ev, ef) if int(v) == x]
    dev.set_messages([x, x], facs[:2], L.TO_VARIABLE, L.FORM_NATURAL, np.array([[0.0, -1.0], [np.inf, 1.0]]))
    broke = dev.message_health()
    assert broke["negative_precision"] == 1 and broke["non_finite"] == 1 and broke["undefined"] == 0 and broke["defined"] == n


def test_loopy_grid_after_seeding(hip_lib):
    m = cx.synth.gaussian_grid(40, 37, seed=5)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, dev, seed_variance=1e6)
    n = _readable(m)
    dev.sweep(5)
    assert dev.message_health() == {"defined": n, "undefined": 0, "negative_precision": 0, "non_finite": 0}


@pytest.mark.parametrize("d,T,schedule", [(4, 300, L.SCHED_CHAIN_SCAN), (2, 50, L.SCHED_TREE), (64, 12, L.SCHED_CHAIN_SCAN), (6, 9, L.SCHED_TREE), (16, 9, L.SCHED_TREE), (20, 12, L.SCHED_CHAIN_SCAN)])
def test_d_dimensional_messages(hip_lib, d, T, schedule):
    m = cx.synth.lgssm_chain(T, d=d, seed=4)
    dev = cx.DeviceGraph(dim=d, schedule=schedule)
    cx.synth.load_into_device(m, dev)
    n = _readable(m)
    before = dev.message_health()
    assert before["defined"] + before["undefined"] == n and before["undefined"] >= 2 * (T - 1)      # the chain messages at least
    dev.sweep(1)
    assert dev.message_health() == {"defined": n, "undefined": 0, "negative_precision": 0, "non_finite": 0}


def test_a_flooding_run_without_seeds_stays_undefined_where_nothing_arrives(hip_lib):
    """the fused schedule on a chain without seeding: information moves one link per sweep — after k sweeps the messages further than k
    links from the data-carrying ends of the dependency chain are still undefined, and the counter says how many"""
    T = 200
    m = cx.synth.ssm_chain(T, seed=9)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, dev)
    n = _readable(m)
    seen = []
    for k in (1, 5, 50, 2 * T):
        dev.sweep(k if not seen else k - sum(seen))
        seen.append(k - sum(seen))
        h = dev.message_health()
        assert h["defined"] + h["undefined"] == n
        if k < T // 2:
            assert h["undefined"] > 0
    assert dev.message_health()["undefined"] == 0


def test_variational_handles_refuse(hip_lib):
    vm = cx.synth.vmp_ssm(8, seed=1)
    dev = cx.DeviceGraph(family=L.FAMILY_VMP_MEAN_FIELD)
    cx.synth.load_vmp_into_device(vm, dev)
    with pytest.raises(cx.CortexHipError, match="not available for the variational families"):
        dev.message_health()
