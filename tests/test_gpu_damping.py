"""-m gpu: cx_set_damping — new = (1 - lambda) rule + lambda old for every factor→variable message of a fused / flooding sweep.

The reference has no damping (its rules are user code); what can be pinned is (i) the arithmetic, sweep by sweep, against a few lines of
numpy that apply the definition to the moment-form checker's sweep; (ii) that lambda = 0 changes nothing, bit for bit; (iii) that the
fixed point is the undamped one wherever that exists; (iv) what damping is for: a frustrated model — positive definite, not
walk-summable — on which loopy Gaussian BP diverges undamped and reaches the exact means damped."""
import itertools

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from tests.helpers import assert_close, flood_oracle_from_model, random_loopy_model

pytestmark = pytest.mark.gpu


def _frustrated_k4(r, seed=1):
    """Four variables, every pair tied by x_j = -x_i + N(0, 1/r) (a repulsive coupling: J_ij = +r), unary messages of natural
    parameters (xi_i, 1 - 3r): J = I + r (11' - I), positive definite for r < 1, walk-summable only for r < 1/3."""
    n = 4
    pairs = list(itertools.combinations(range(n), 2))
    x = np.arange(1, n + 1, dtype=np.int64)
    unary = x + n
    pf = 2 * n + 1 + np.arange(len(pairs), dtype=np.int64)
    pa = np.array([p[0] for p in pairs]) + 1
    pb = np.array([p[1] for p in pairs]) + 1
    params = np.zeros((n + len(pairs), 3))
    params[n:, 0], params[n:, 1] = 1.0 / r, -1.0
    role = np.concatenate([np.full(n, L.ROLE_OUT), np.full(len(pairs), L.ROLE_IN), np.full(len(pairs), L.ROLE_OUT)]).astype(np.int32)
    xi = np.random.default_rng(seed).standard_normal(n)
    model = cx.synth.Model(edge_var=np.concatenate([x, pa, pb]), edge_fac=np.concatenate([unary, pf, pf]), factor_ids=np.concatenate([unary, pf]),
                           factor_kind=np.concatenate([np.zeros(n, np.int32), np.full(len(pairs), L.FACTOR_GAUSS_LINEAR, np.int32)]),
                           factor_var=params, x_ids=x, edge_role=role)
    J = np.eye(n) + r * (np.ones((n, n)) - np.eye(n))
    return model, unary, xi, np.full(n, 1.0 - 3.0 * r), np.linalg.solve(J, xi)


def _load_k4(model, unary, xi, w, schedule):
    dev = cx.DeviceGraph(schedule=schedule)
    cx.synth.load_into_device(model, dev)
    dev.set_messages(model.x_ids, unary, L.TO_VARIABLE, L.FORM_NATURAL, np.stack([xi, w], axis=1))
    dev.seed_messages(L.TO_VARIABLE, 0.0, 1e6)
    return dev


@pytest.mark.parametrize("schedule", [L.SCHED_FUSED, L.SCHED_FLOODING])
def test_damping_rescues_a_frustrated_model(hip_lib, schedule):
    model, unary, xi, w, want = _frustrated_k4(0.34)
    dev = _load_k4(model, unary, xi, w, schedule)
    dev.sweep(1500)
    got = dev.get_marginals(model.x_ids)[:, 0]
    assert not np.all(np.isfinite(got)) or np.max(np.abs(got - want)) > 1.0, "undamped loopy BP diverges on this model"
    dev.close()
    dev = _load_k4(model, unary, xi, w, schedule)
    dev.set_damping(0.3)
    dev.sweep(400)
    assert_close(dev.get_marginals(model.x_ids)[:, 0], want, 1e-9, "damped: the exact means", scale_by="max")
    dev.close()


def _numpy_damped_sweeps(g, lam, n):
    """the moment-form checker's flooding sweep (oracle/bp_flood.c), damped by the definition: natural parameters mixed"""
    for _ in range(n):
        old_m, old_v = g.f2v_m.copy(), g.f2v_v.copy()
        g.sweep(1)
        upd = g.partner >= 0
        xi_old, w_old = old_m[upd] / old_v[upd], 1.0 / old_v[upd]
        xi_new, w_new = g.f2v_m[upd] / g.f2v_v[upd], 1.0 / g.f2v_v[upd]
        xi, w = (1 - lam) * xi_new + lam * xi_old, (1 - lam) * w_new + lam * w_old
        g.f2v_m[upd], g.f2v_v[upd] = xi / w, 1.0 / w


@pytest.mark.parametrize("schedule", [L.SCHED_FUSED, L.SCHED_FLOODING])
@pytest.mark.parametrize("name", ["grid", "hubs"])
def test_damped_sweeps_against_the_definition_sweep_by_sweep(hip_lib, schedule, name):
    """(hubs: variables of degree > 8 — the CSR tail's part of a fused sweep is damped like the rest)"""
    model = cx.synth.gaussian_grid(24, 19, seed=3) if name == "grid" else random_loopy_model(5, 1, nv=40, extra=70)[0]
    lam = 0.35
    dev = cx.DeviceGraph(schedule=schedule)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.set_damping(lam)
    g = flood_oracle_from_model(model, seed_variance=1e6)
    for k in range(6):
        dev.sweep(1)
        _numpy_damped_sweeps(g, lam, 1)
        got = dev.get_messages(g.edge_var, g.edge_fac, L.TO_VARIABLE)
        assert_close(got[:, 0], g.f2v_m, 1e-9, f"sweep {k + 1} f2v mean")
        assert_close(got[:, 1], g.f2v_v, 1e-9, f"sweep {k + 1} f2v variance")
    dev.close()


def test_zero_is_off_bit_for_bit_and_the_fixed_point_does_not_move(hip_lib):
    model = cx.synth.gaussian_grid(24, 19, seed=3)
    plain = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, plain, seed_variance=1e6)
    zero = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, zero, seed_variance=1e6)
    zero.set_damping(0.4); zero.set_damping(0.0)
    plain.sweep(7); zero.sweep(7)
    ev, ef = model.edge_var, model.edge_fac
    assert np.array_equal(plain.get_messages(ev, ef, L.TO_VARIABLE), zero.get_messages(ev, ef, L.TO_VARIABLE))
    damp = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, damp, seed_variance=1e6)
    damp.set_damping(0.5)
    plain.sweep(400); damp.sweep(900)
    assert_close(damp.get_messages(ev, ef, L.TO_VARIABLE), plain.get_messages(ev, ef, L.TO_VARIABLE), 1e-9, "the damped iteration's fixed point")
    me = exact.grid_posterior_mean(24, 19, model.meta["y"], model.meta["r"], model.meta["qh"], model.meta["qv"])
    assert_close(damp.get_marginals(model.x_ids)[:, 0], me, 1e-8, "converged means vs the sparse solve")
    for d in (plain, zero, damp):
        d.close()


@pytest.mark.parametrize("d", [2, 4, 64, 9])
def test_d_dimensional_damped_sweeps_keep_the_fixed_point(hip_lib, d):
    """dim 2..4 — and (round 6) dim 64, 5 .. 63 embedded in it: natural parameters (eta, Lambda) mixed entry by entry; on a tree the fixed
    point is the exact posterior"""
    from tests.test_gpu_mv import _branching_lgssm

    nst = 31 if d <= 4 else 15
    model, emean, ecov = _branching_lgssm(nst, d, seed=40 + d, b=2)
    dev = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.set_damping(0.4)
    dev.sweep(260 if d <= 4 else 150)
    marg = dev.get_marginals(model.x_ids)
    assert_close(marg[:, :d], emean, 1e-8, "damped d-dimensional sweeps: marginal mean", scale_by="max")
    assert_close(marg[:, d:].reshape(nst, d, d), ecov, 1e-8, "marginal covariance", scale_by="max")
    # one damped sweep against the definition: mix the undamped result with the previous message in natural form
    a = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED); b = cx.DeviceGraph(dim=d, schedule=L.SCHED_FUSED)
    for h in (a, b):
        cx.synth.load_into_device(model, h, seed_variance=1e6)
        h.sweep(3)
    xs = set(int(v) for v in model.x_ids)
    keep = np.array([int(v) in xs for v in model.edge_var])
    ev, ef = model.edge_var[keep], model.edge_fac[keep]
    old = a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    a.sweep(1)
    new = a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    b.set_damping(0.25)
    b.sweep(1)
    got = b.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    want = np.where(np.isnan(old), new, 0.75 * new + 0.25 * old)
    assert_close(got, want, 1e-10, "one damped sweep, natural parameters", scale_by="max")
    for h in (dev, a, b):
        h.close()


def test_refusals(hip_lib):
    model = cx.synth.ssm_chain(20, seed=1)
    for sched in (L.SCHED_CHAIN_SCAN, L.SCHED_TREE, L.SCHED_REFERENCE):
        dev = cx.DeviceGraph(schedule=sched)
        cx.synth.load_into_device(model, dev)
        with pytest.raises(cx.CortexHipError) as ei:
            dev.set_damping(0.2)
        assert ei.value.code == L.ERR_UNSUPPORTED
        dev.set_damping(0.0)
        dev.close()
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev)
    for bad in (-0.1, 1.0, float("nan")):
        with pytest.raises(cx.CortexHipError) as ei:
            dev.set_damping(bad)
        assert ei.value.code == L.ERR_INVALID_ARGUMENT
    dev.close()


def test_message_halos_and_damping_exclude_each_other_in_either_order(hip_lib):
    """ cx_set_damping refuses a handle with per-sweep message halos; cx_halo_configure must refuse a damped handle too"""
    from cortex.jl_amd import partition
    part = partition.grid_rows(12, 10, 0, 2, seed=3)
    a = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(part.model, a, seed_variance=1e6)
    a.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
    with pytest.raises(cx.CortexHipError) as ei:
        a.set_damping(0.5)
    assert ei.value.code == L.ERR_UNSUPPORTED
    b = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(part.model, b, seed_variance=1e6)
    b.set_damping(0.5)
    with pytest.raises(cx.CortexHipError, match="not damped") as ei:
        b.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
    assert ei.value.code == L.ERR_UNSUPPORTED
    b.halo_configure([], [], [], [])          # taking a halo away is always possible
    b.set_damping(0.0)
    b.halo_configure(part.send_var, part.send_fac, part.recv_var, part.recv_fac)
    # state halos run plain sweeps and are damped like them
    c = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    deep = partition.grid_rows_deep(12, 10, 0, 2, 2, seed=3)
    cx.synth.load_into_device(deep.model, c, seed_variance=1e6)
    c.set_damping(0.3)
    c.halo_configure_state(deep.send_var, deep.send_fac, deep.recv_var, deep.recv_fac)
    for d in (a, b, c):
        d.close()
