"""-m gpu: linear-Gaussian factors with more than two edges (CX_FACTOR_GAUSS_LINEAR_N, csrc/cx_kary.hip).

The reference wires every factor→variable message of a factor to ALL the other variable→factor messages of that factor
(src/dependencies.jl:17-31) and leaves the rule to the user; the device's rule is checked per sweep against oracle/bp_kary.c (the
same flooding sweep in moment form, pinned by a dense solve in tests/test_kary_checker.py), on trees against the dense posterior
itself, and message by message through cx_update_batch."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import ref
from tests.helpers import assert_close
from tests.kary_support import dense_posterior

pytestmark = pytest.mark.gpu


def _checker(m, seed_variance=None):
    g = ref.KaryFloodGraph(m)
    g.set_message_to_variable(m.prior_var, m.prior_fac, m.prior_mean, m.prior_variance)
    if len(m.data_var):
        g.set_data(m.data_var, m.data_fac, m.data_y)
    if seed_variance is not None:
        und = np.isnan(g.f2v_v) & g.kary_edge
        g.f2v_m[und], g.f2v_v[und] = 0.0, seed_variance
    return g


@pytest.mark.parametrize("schedule", [L.SCHED_FUSED, L.SCHED_FLOODING])
@pytest.mark.parametrize("n_factors,tree,seed,observe", [(1, True, 1, 0.0), (30, True, 2, 0.3), (200, False, 3, 0.0), (3000, False, 4, 0.0), (5000, True, 5, 0.2)])
def test_every_sweep_equals_the_checker(hip_lib, schedule, n_factors, tree, seed, observe):
    """trees start from undefined messages (definedness spreads exactly as in the checker), loopy graphs from a vague seed"""
    m = cx.synth.kary_model(n_factors, seed=seed, tree=tree, observe=observe)
    sv = None if tree else 1e3
    dev = cx.DeviceGraph(schedule=schedule)
    cx.synth.load_into_device(m, dev, seed_variance=sv)
    g = _checker(m, sv)
    for sweep in range(6):
        dev.sweep(1)
        g.sweep(1)
        got = dev.get_messages(g.edge_var, g.edge_fac, L.TO_VARIABLE)
        assert_close(got[:, 0], g.f2v_m, 1e-9, f"sweep {sweep}: means of {g.ne} factor→variable messages")
        assert_close(got[:, 1], g.f2v_v, 1e-9, f"sweep {sweep}: variances")
    assert dev.stats()["n_messages_per_sweep"] >= int(g.kary_edge.sum())


@pytest.mark.parametrize("schedule", [L.SCHED_FUSED, L.SCHED_FLOODING])
def test_tree_fixed_point_is_the_exact_posterior(hip_lib, schedule):
    m = cx.synth.kary_model(60, seed=11, tree=True, observe=0.25)
    dev = cx.DeviceGraph(schedule=schedule)
    cx.synth.load_into_device(m, dev)
    dev.sweep(2 * 60 + 6)
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "marginal means vs the dense solve")
    assert_close(marg[:, 1], ev, 1e-9, "marginal variances vs the dense solve")


def test_messages_one_by_one_through_update_batch(hip_lib):
    """cx_update_batch: MessageToFactor, MessageToVariable (of a k-ary factor: from ALL its other stored messages) and IndividualMarginal
    items in the order a sequential scheduler would issue them on a one-factor tree == the fixed point of the sweeps"""
    m = cx.synth.kary_model(1, seed=21, k_choices=(5,))
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, dev)
    fid = int(m.meta["kary_ids"][0])
    vs = [int(v) for v in m.meta["fac_vars"][0]]
    n = len(vs)
    dev.update_batch([L.ITEM_MESSAGE_TO_FACTOR] * n, vs, [fid] * n)                 # each = its variable's prior
    dev.update_batch([L.ITEM_MESSAGE_TO_VARIABLE] * n, vs, [fid] * n)               # each reads the five others
    dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * n, vs, [0] * n)
    ids, em, ev = dense_posterior(m)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "batched: marginal means")
    assert_close(marg[:, 1], ev, 1e-9, "batched: marginal variances")


def test_new_coefficients_and_checkpoint_fingerprint(hip_lib):
    m = cx.synth.kary_model(8, seed=31, tree=True)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, dev)
    dev.sweep(24)
    blob = dev.export_state()
    m2 = cx.synth.kary_model(8, seed=31, tree=True)
    m2.meta["coef"] = m2.meta["coef"] * 1.5
    dev.set_factor_coefficients(m2.meta["coef_var"], m2.meta["coef_fac"], m2.meta["coef"])
    dev.sweep(24)
    ids, em, ev = dense_posterior(m2)
    marg = dev.get_marginals(ids)
    assert_close(marg[:, 0], em, 1e-9, "after new coefficients: means")
    assert_close(marg[:, 1], ev, 1e-9, "after new coefficients: variances")
    with pytest.raises(cx.CortexHipError):            # the blob belongs to the old coefficients
        dev.import_state(blob)
    other = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, other)
    other.import_state(blob)
    ids, em, ev = dense_posterior(m)
    assert_close(other.get_marginals(ids)[:, 0], em, 1e-9, "imported state: means")


def test_refusals(hip_lib):
    m = cx.synth.kary_model(2, seed=41, k_choices=(3,))
    with pytest.raises(cx.CortexHipError, match="not a link of a chain"):
        cx.synth.load_into_device(m, cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN))
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    with pytest.raises(cx.CortexHipError, match="edge roles"):
        dev.graph_create(m.edge_var, m.edge_fac, m.factor_ids, m.factor_kind, m.factor_var)
    bad = m.edge_role.copy()
    bad[:] = L.ROLE_IN
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    with pytest.raises(cx.CortexHipError, match="CX_ROLE_OUT"):
        dev.graph_create(m.edge_var, m.edge_fac, m.factor_ids, m.factor_kind, m.factor_var, edge_role=bad)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(m, dev)
    with pytest.raises(cx.CortexHipError, match="not a ROLE_IN edge"):
        dev.set_factor_coefficients(m.prior_var[:1], m.prior_fac[:1], [2.0])
    with pytest.raises(cx.CortexHipError, match="non-zero"):
        dev.set_factor_coefficients(m.meta["coef_var"][:1], m.meta["coef_fac"][:1], [0.0])
    dev.halo_configure_state([], [], [], [])           # (round 5) state halos take such graphs: tests/test_gpu_partition.py
    with pytest.raises(cx.CortexHipError, match="unary and pairwise"):
        dev.halo_configure([], [], [], [])             # the per-sweep message halo does not
