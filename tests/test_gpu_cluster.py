"""-m gpu: the XCD-resident cluster (cx_batch.hip: k_ref_cluster) against the same plans run as launches.

Reference-order plans of many dependent stages of 1 - 16 k items — a call on a large loopy grid — run as ONE launch of the workgroups of one XCD behind barriers that stay in that XCD's L2, with values loaded past the vector cache, flat
records and helper workgroups that load ahead.  The plan and its items are the same either way; what is pinned here is that the cluster
computes what the launches compute (the sums of a compact MessageToFactor item are associated left to right on the cluster, prefix + suffix
as launches: equal to rounding), on the paths that choose it by themselves, and that the parity tests of the other files therefore cover it."""
import os

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def _pair(make):
    """the same handle twice: on the cluster, and with CX_REF_CLUSTER=0 (read when the handle prepares its cluster)"""
    a = make()
    os.environ["CX_REF_CLUSTER"] = "0"
    try:
        b = make()
    finally:
        del os.environ["CX_REF_CLUSTER"]
    return a, b


def test_a_reference_order_call_on_a_grid_runs_on_the_cluster(hip_lib):
    model = cx.synth.gaussian_grid(1200, 1300, seed=3)      # stages of a median ~ 1,400 items: wider than one workgroup's run, narrower than the cluster
    prior = np.stack([model.prior_mean, model.prior_variance], axis=1)

    def make():
        dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        return dev

    a, b = _pair(make)
    for call in range(3):
        for dev in (a, b):
            dev.sweep(1)
        assert a.ref_trace() == b.ref_trace()
        sa, sb = a.cluster_stats(), b.cluster_stats()
        assert sa["state"] == 1 and sa["last_reference_call"] and sa["workgroups"] >= 32, sa
        assert sb["state"] == -1 and not sb["last_reference_call"], sb
        assert a.ref_plan_stats()["launches"] <= 4 < b.ref_plan_stats()["launches"]      # (the two stages that hold every prior at once leave as launches either way)
        for direction in (L.TO_VARIABLE, L.TO_FACTOR):
            ma, mb = a.get_messages(model.edge_var, model.edge_fac, direction, L.FORM_NATURAL), b.get_messages(model.edge_var, model.edge_fac, direction, L.FORM_NATURAL)
            assert np.array_equal(np.isnan(ma), np.isnan(mb))
            ok = ~np.isnan(ma)
            assert_close(ma[ok], mb[ok], 1e-12, f"call {call}: messages, direction {direction}", scale_by="max")
        assert_close(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids), 1e-12, f"call {call}: marginals", scale_by="max")
        for dev in (a, b):
            dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)


def test_two_cluster_handles_agree_in_every_bit(hip_lib):
    """the same plan on two handles: whichever XCD and workgroups a launch lands on, the arithmetic per execution is the same, so every stored
    value agrees bit for bit after every call — a value read before its writer's store had reached the L2 would show here (the long form:
    tools/lab/soak_cluster.py, 150 calls at C4)"""
    model = cx.synth.gaussian_grid(900, 1000, seed=5)
    prior = np.stack([model.prior_mean, model.prior_variance], axis=1)
    devs = []
    for _ in range(2):
        dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        devs.append(dev)
    for call in range(6):
        for dev in devs:
            dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
            dev.sweep(1)
        assert all(d.cluster_stats()["last_reference_call"] for d in devs)
        a, b = (d.get_marginals(model.x_ids) for d in devs)
        assert np.array_equal(a, b), f"call {call}: marginals"
        for direction in (L.TO_VARIABLE, L.TO_FACTOR):
            ma, mb = (d.get_messages(model.edge_var, model.edge_fac, direction, L.FORM_NATURAL) for d in devs)
            assert np.array_equal(ma, mb, equal_nan=True), f"call {call}: messages, direction {direction}"
    for d in devs:
        d.close()


def test_a_small_grid_stays_with_one_workgroups_runs(hip_lib):
    """stages of at most 1,024 items fold into runs of one workgroup with its own barrier (≈ 1 us a stage): cheaper than the cluster's"""
    model = cx.synth.gaussian_grid(100, 110, seed=3)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(1)
    assert dev.cluster_stats()["state"] == 1 and not dev.cluster_stats()["last_reference_call"]


def test_plans_of_thin_stages_and_short_plans_stay_with_launches(hip_lib, monkeypatch):
    monkeypatch.setenv("CX_REF_CHAIN_MIN", "0")      # (the chains of this plan as stages, not as the scan steps of round 6: tests/test_gpu_reference_schedule.py)
    chain = cx.synth.ssm_chain(400, seed=1)
    dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(chain, dev)
    dev.sweep_for(chain.x_ids)
    st = dev.cluster_stats()
    assert st["state"] == 1 and not st["last_reference_call"], st      # 400 stages of a few items: one workgroup's own barrier (k_batch_run)
    assert dev.ref_plan_stats()["launches"] == 1
    monkeypatch.delenv("CX_REF_CHAIN_MIN")
    scans = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
    cx.synth.load_into_device(chain, scans)
    scans.sweep_for(chain.x_ids)
    st = scans.ref_plan_stats()
    assert st["stages"] <= 4 and st["launches"] <= 6, st              # the same call with its two chains as scan steps
    assert_close(scans.get_marginals(chain.x_ids), dev.get_marginals(chain.x_ids), 1e-10, "chains as scan steps == chains as stages", scale_by="max")


@pytest.mark.parametrize("fault_stage", [1, 700])
def test_a_member_that_never_arrives_is_survived(hip_lib, monkeypatch, fault_stage):
    """Fault injection: one member workgroup skips its arrival at one barrier — what a workgroup that never became resident
    (a CU mask, a restricted queue, a third tenant's long kernels) looks like to the others.  Every wait of the cluster is bounded in TIME
    (shortened here), the host finds the first incomplete stage from the arrival counter and finishes the call on plain launches from
    there: the call succeeds with the launches' results, the handle reports one recovered call and stays with launches."""
    model = cx.synth.gaussian_grid(1200, 1300, seed=3)
    prior = np.stack([model.prior_mean, model.prior_variance], axis=1)

    def make():
        dev = cx.DeviceGraph(schedule=L.SCHED_REFERENCE)
        cx.synth.load_into_device(model, dev, seed_variance=1e6)
        return dev

    a, b = _pair(make)
    for dev in (a, b):          # a first, undisturbed call (it also finds the plan)
        dev.sweep(1)
        dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
    assert a.cluster_stats()["state"] == 1 and a.cluster_stats()["last_reference_call"]
    monkeypatch.setenv("CX_REF_CLUSTER_FAULT", str(fault_stage))
    monkeypatch.setenv("CX_REF_CLUSTER_TIMEOUT_MS", "50")
    import time
    t0 = time.perf_counter()
    a.sweep(1)                  # does NOT raise
    a.sync()
    dt = time.perf_counter() - t0
    monkeypatch.delenv("CX_REF_CLUSTER_FAULT")
    b.sweep(1)
    st = a.cluster_stats()
    assert st["state"] == -1 and st["recovered_calls"] == 1, st
    assert dt < 5.0, f"the bounded waits took {dt:.2f} s"
    for call in range(2):       # the faulted call, then one more on the handle that went back to launches
        assert a.ref_trace() == b.ref_trace()
        for direction in (L.TO_VARIABLE, L.TO_FACTOR):
            ma, mb = a.get_messages(model.edge_var, model.edge_fac, direction, L.FORM_NATURAL), b.get_messages(model.edge_var, model.edge_fac, direction, L.FORM_NATURAL)
            assert np.array_equal(np.isnan(ma), np.isnan(mb))
            ok = ~np.isnan(ma)
            assert_close(ma[ok], mb[ok], 1e-12, f"call {call}: messages, direction {direction}", scale_by="max")
        assert_close(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids), 1e-12, f"call {call}: marginals", scale_by="max")
        for dev in (a, b):
            dev.set_messages(model.prior_var, model.prior_fac, L.TO_VARIABLE, L.FORM_MOMENT, prior)
            dev.sweep(1)
    assert a.cluster_stats()["recovered_calls"] == 1 and not a.cluster_stats()["last_reference_call"]
