"""-m gpu: the HIP sweep (through the C ABI) against the CPU checker on the same seeded inputs.

Tolerance: BASELINE.json asks for marginals within 1e-6 relative of the CPU reference path.  The device
computes in natural (information) form, the checker in the reference's moment form, so results differ by
rounding only; these tests hold the device to RTOL = 1e-9 per sweep and 1e-6 at the full sizes."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from tests.helpers import assert_close, engine_oracle_from_model, flood_oracle_from_model

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _device(model, schedule, seed_variance=None, materialize=False):
    dev = cx.DeviceGraph(schedule=schedule, materialize_messages_to_factor=materialize)
    cx.synth.load_into_device(model, dev, seed_variance)
    return dev


def _compare_messages(dev, g, what):
    both = g.partner >= 0
    f2v = dev.get_messages(g.edge_var, g.edge_fac, L.TO_VARIABLE)
    assert_close(f2v[:, 0], g.f2v_m, RTOL, what + " f2v mean")
    assert_close(f2v[:, 1], g.f2v_v, RTOL, what + " f2v variance")
    v2f = dev.get_messages(g.edge_var[both], g.edge_fac[both], L.TO_FACTOR)
    assert_close(v2f[:, 0], g.v2f_m[both], RTOL, what + " v2f mean")
    assert_close(v2f[:, 1], g.v2f_v[both], RTOL, what + " v2f variance")


SCHEDULES = [(L.SCHED_FLOODING, False), (L.SCHED_FUSED, False), (L.SCHED_FUSED, True)]


@pytest.mark.parametrize("schedule,materialize", SCHEDULES)
@pytest.mark.parametrize("shape", [(1, 2), (2, 2), (3, 7), (16, 16), (37, 23), (64, 300)])
def test_grid_flooding_sweeps_match_oracle(hip_lib, schedule, materialize, shape):
    model = cx.synth.gaussian_grid(*shape, seed=7)
    dev = _device(model, schedule, seed_variance=1e6, materialize=materialize)
    g = flood_oracle_from_model(model, seed_variance=1e6)
    st = dev.stats()
    assert st["n_edges"] == model.n_edges == g.ne
    for sweep in range(6):
        dev.sweep(1)
        n_upd = g.sweep(1)
        assert n_upd == st["n_messages_per_sweep"]
        _compare_messages(dev, g, f"grid{shape} sweep {sweep}")
        m, v = g.marginals()
        # marginals written by the sweep are those of the messages the sweep STARTED from
        # → compare after one more variable phase via the batch path instead
    dev.sweep(40)
    g.sweep(40)
    _compare_messages(dev, g, f"grid{shape} after 46 sweeps")
    # BP means at convergence are the exact posterior means (loopy Gaussian BP), schedule-independent
    for _ in range(30):
        dev.sweep(100)
        if dev.residual() < 1e-13:
            break
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    mean_exact = exact.grid_posterior_mean(shape[0], shape[1], model.meta["y"], model.meta["r"], model.meta["qh"], model.meta["qv"])
    assert_close(marg[:, 0], mean_exact, 1e-8, f"grid{shape} converged mean vs sparse solve")


@pytest.mark.parametrize("schedule,materialize", SCHEDULES)
@pytest.mark.parametrize("T,randvar", [(2, False), (3, False), (50, False), (200, True), (700, True)])
def test_chain_flooding_reaches_reference_marginals(hip_lib, schedule, materialize, T, randvar):
    """On a tree the flooding fixed point equals what the reference's sequential schedule computes in one
    update_marginals! call, and both equal the exact smoother."""
    model = cx.synth.ssm_chain(T, seed=3, random_variances=randvar)
    dev = _device(model, schedule, materialize=materialize)
    g = flood_oracle_from_model(model)
    for sweep in range(T + 2):
        dev.sweep(1)
        g.sweep(1)
        if sweep < 4 or sweep == T + 1:
            _compare_messages(dev, g, f"chain T={T} sweep {sweep}")
    dev.sweep(1)  # marginal refresh from the converged messages
    marg = dev.get_marginals(model.x_ids)
    E = engine_oracle_from_model(model)
    E.update_marginals(model.x_ids)
    _, em, ev = E.get_marginals(model.x_ids)
    assert_close(marg[:, 0], em, RTOL, "marginal mean vs restated reference scheduler")
    assert_close(marg[:, 1], ev, RTOL, "marginal variance vs restated reference scheduler")
    xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
    assert_close(marg[:, 0], xm, 1e-9, "marginal mean vs tridiagonal solve")
    assert_close(marg[:, 1], xv, 1e-9, "marginal variance vs tridiagonal solve")


def test_batch_mode_replays_reference_schedule(hip_lib):
    """Drive the device one reference `process!` at a time, in the exact order the restated scheduler
    executes (schedule/indexing parity), and get the same messages and marginals."""
    T = 40
    model = cx.synth.ssm_chain(T, seed=11, random_variances=True)
    dev = _device(model, L.SCHED_FLOODING)
    E = engine_oracle_from_model(model, trace=True)
    E.update_marginals(model.x_ids)
    kinds, vs, fs = [], [], []
    for _round, _vid, sig, _before, _after in E.trace():
        k, var, fac, _, _ = E.variant(sig)
        kinds.append({1: L.ITEM_MESSAGE_TO_FACTOR, 2: L.ITEM_MESSAGE_TO_VARIABLE, 4: L.ITEM_INDIVIDUAL_MARGINAL}[k])
        vs.append(var)
        fs.append(fac)
    assert len(kinds) == 5 * T - 4 + T
    for k, v, f in zip(kinds, vs, fs):
        dev.update_batch([k], [v], [f])
    marg = dev.get_marginals(model.x_ids)
    _, em, ev = E.get_marginals(model.x_ids)
    assert_close(marg[:, 0], em, RTOL, "batched marginal mean")
    assert_close(marg[:, 1], ev, RTOL, "batched marginal variance")
    tr = model.factor_ids[T:]
    _, mm, mv = E.get_messages(model.x_ids[:-1], tr, to_variable=True)
    got = dev.get_messages(model.x_ids[:-1], tr, L.TO_VARIABLE)
    assert_close(got[:, 0], mm, RTOL, "backward message mean")
    assert_close(got[:, 1], mv, RTOL, "backward message variance")


def test_errors_are_statuses_not_crashes(hip_lib):
    dev = cx.DeviceGraph()
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)
    assert e.value.code == L.ERR_STATE
    model = cx.synth.ssm_chain(4)
    cx.synth.load_into_device(model, dev)
    with pytest.raises(cx.CortexHipError) as e:
        dev.get_messages([1], [999], L.TO_VARIABLE)
    assert e.value.code == L.ERR_NOT_FOUND and "999" in e.value.message
    with pytest.raises(cx.CortexHipError) as e:
        dev.update_batch([3], [1], [1])  # 3 is no item kind (CX_ITEM_* are 1, 2, 4, 8, 16): the reference's error("...no rule...") branch
    assert e.value.code == L.ERR_UNSUPPORTED


def _random_pairwise_graph(rng, n_var, n_pair, hubs=()):
    """variables 1..n_var with a unary prior each, random pairwise factors, plus hub variables of given degrees."""
    ev, ef, fids, kinds, fq = [], [], [], [], []
    nid = n_var
    prior_var, prior_fac = [], []
    for v in range(1, n_var + 1):
        nid += 1
        ev.append(v); ef.append(nid); fids.append(nid); kinds.append(L.FACTOR_OPAQUE); fq.append(1.0)
        prior_var.append(v); prior_fac.append(nid)
    pairs = set()
    while len(pairs) < n_pair:
        a, b = rng.integers(1, n_var + 1, 2)
        if a != b:
            pairs.add((int(min(a, b)), int(max(a, b))))
    for hub, deg in hubs:
        others = rng.choice(np.setdiff1d(np.arange(1, n_var + 1), [hub]), size=deg, replace=False)
        for o in others:
            pairs.add((int(min(hub, o)), int(max(hub, o))))
    for a, b in sorted(pairs):
        nid += 1
        ev += [a, b]; ef += [nid, nid]; fids.append(nid); kinds.append(L.FACTOR_GAUSS_ADDITIVE); fq.append(float(rng.uniform(0.5, 2.0)))
    pm, pv = rng.standard_normal(n_var), rng.uniform(0.5, 2.0, n_var)
    return cx.synth.Model(edge_var=np.array(ev), edge_fac=np.array(ef), factor_ids=np.array(fids),
                          factor_kind=np.array(kinds, dtype=np.int32), factor_var=np.array(fq), x_ids=np.arange(1, n_var + 1),
                          prior_var=np.array(prior_var), prior_fac=np.array(prior_fac), prior_mean=pm, prior_variance=pv)


@pytest.mark.parametrize("schedule,materialize", SCHEDULES)
def test_ragged_degrees_and_high_degree_variables(hip_lib, schedule, materialize):
    """degrees 1..8 take the SELL path, hubs of degree 9, 40, 300 and 1500 the wave-per-variable scans
    (the device analogue of the reference's segment tree, dependencies.jl:90-173)."""
    rng = np.random.default_rng(42)
    model = _random_pairwise_graph(rng, 2000, 2500, hubs=[(7, 9), (300, 40), (999, 300), (1500, 1500)])
    dev = _device(model, schedule, seed_variance=50.0, materialize=materialize)
    g = flood_oracle_from_model(model, seed_variance=50.0)
    assert dev.stats()["n_big_variables"] >= 4
    for sweep in range(5):
        dev.sweep(1)
        assert g.sweep(1) == dev.stats()["n_messages_per_sweep"]
        _compare_messages(dev, g, f"ragged sweep {sweep}")
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    m, v = g.marginals()
    assert_close(marg[:, 0], m, RTOL, "ragged marginal mean")
    assert_close(marg[:, 1], v, RTOL, "ragged marginal variance")


def test_empty_batches_and_lists_are_noops(hip_lib):
    model = cx.synth.ssm_chain(6)
    dev = _device(model, L.SCHED_FUSED)
    dev.update_batch([], [], [])
    dev.set_messages([], [], L.TO_FACTOR, L.FORM_POINT, [])
    assert dev.get_messages([], [], L.TO_VARIABLE).shape == (0, 2)
    assert dev.get_marginals([]).shape == (0, 2)
    dev.sweep(0)
    assert dev.stats()["sweeps_done"] == 0


def test_config_c4_full_size(hip_lib):
    """BASELINE.json's metric config: the 10,005,465-edge Gaussian grid, at full size.
    (i) three sweeps against the CPU checker on 300k sampled messages and all marginals;
    (ii) size-independent property: at convergence the BP means solve J m = h of the grid's precision matrix."""
    N = 1415
    model = cx.synth.gaussian_grid(N, N, seed=1234)
    assert model.n_edges == 10_005_465
    dev = _device(model, L.SCHED_FUSED, seed_variance=1e6)
    st = dev.stats()
    assert st["n_edges"] == 10_005_465 and st["n_messages_per_sweep"] == 16_006_480
    g = flood_oracle_from_model(model, seed_variance=1e6)
    dev.sweep(3)
    assert g.sweep(3, use_omp=True) == 3 * 16_006_480
    rng = np.random.default_rng(0)
    pick = rng.choice(np.flatnonzero(g.partner >= 0), size=300_000, replace=False)
    got = dev.get_messages(g.edge_var[pick], g.edge_fac[pick], L.TO_VARIABLE)
    assert_close(got[:, 0], g.f2v_m[pick], RTOL, "C4 f2v mean (sampled)")
    assert_close(got[:, 1], g.f2v_v[pick], RTOL, "C4 f2v variance (sampled)")
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    m, v = g.marginals(use_omp=True)
    assert_close(marg[:, 0], m, RTOL, "C4 marginal mean")
    assert_close(marg[:, 1], v, RTOL, "C4 marginal variance")
    dev.residual()
    for _ in range(12):
        dev.sweep(100)
        if dev.residual() < 1e-12:
            break
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    J = exact.grid_precision(N, N, model.meta["r"], model.meta["qh"], model.meta["qv"])
    h = (model.meta["y"] / model.meta["r"]).ravel()
    resid = np.abs(J @ marg[:, 0] - h).max() / np.abs(h).max()
    assert resid < 1e-9, f"converged BP means do not solve the normal equations: relative residual {resid:.2e}"


@pytest.mark.parametrize("schedule", [L.SCHED_FLOODING, L.SCHED_FUSED, L.SCHED_TREE])
@pytest.mark.parametrize("n", [1, 3, 5, 8, 9, 100, 3000])
def test_beta_bernoulli_known_answer_on_device(hip_lib, schedule, n):
    """The reference's conjugate known answer (test/inference_engine_tests.jl:360-376): posterior Beta(1 + Σ, 1 + n − Σ)
    of the star-shaped Beta-Bernoulli model, computed by the device's product-of-messages path (SELL leave-one-out for
    n ≤ 8, wave-scan "segment tree" for n > 8) on messages of the generic 2-parameter family: Beta(a, b) travels as its
    natural parameters (a − 1, b − 1), so the reference's Beta product (a + a' − 1, b + b' − 1) is a plain sum."""
    rng = np.random.default_rng(n)
    data = rng.random(n) < 0.5
    p = 1
    o = 2 * np.arange(1, n + 1)          # ids as make_beta_bernoulli_model hands them out: o_i, f_i alternate (:311-327)
    f = o + 1
    dev = cx.DeviceGraph(schedule=schedule, family=L.FAMILY_NATURAL2)
    dev.graph_create(np.concatenate([np.full(n, p), o]), np.concatenate([f, f]), f, np.full(n, L.FACTOR_OPAQUE, np.int32), np.ones(n))
    # compute_message_to_variable! of the reference's processor: Beta(1 + r, 2 − r) (:256-258) → natural (r, 1 − r)
    r = data.astype(float)
    dev.set_messages(np.full(n, p), f, L.TO_VARIABLE, L.FORM_NATURAL, np.stack([r, 1.0 - r], axis=1))
    dev.sweep(1)
    nat = dev.get_marginals([p])[0]
    assert (1.0 + nat[0], 1.0 + nat[1]) == (1.0 + data.sum(), 1.0 + n - data.sum())   # exact: small integers in f64
    dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL], [p], [0])
    nat = dev.get_marginals([p])[0]
    assert (1.0 + nat[0], 1.0 + nat[1]) == (1.0 + data.sum(), 1.0 + n - data.sum())
    if n >= 2:   # "product of all but me": the message towards factor f_1 is the sum of the others
        dev.update_batch([L.ITEM_MESSAGE_TO_FACTOR], [p], [int(f[0])])
        m = dev.get_messages([p], [int(f[0])], L.TO_FACTOR, L.FORM_NATURAL)[0]
        assert (m[0], m[1]) == (r[1:].sum(), (1.0 - r[1:]).sum())
    with pytest.raises(cx.CortexHipError):
        dev.get_messages([p], [int(f[0])], L.TO_VARIABLE, L.FORM_MOMENT)   # moment form is Gaussian-only


def test_sweep_until_stops_at_the_requested_residual(hip_lib):
    """cx_sweep_until on a loopy grid: stops once the largest message change over `check_every` sweeps is below tol, well
    before max_sweeps; the marginal means then equal the dense solve (loopy Gaussian BP means are exact at convergence)."""
    model = cx.synth.gaussian_grid(24, 31, seed=2)
    dev = _device(model, L.SCHED_FUSED, seed_variance=1e6)
    n, r = dev.sweep_until(1e-12, 5000, check_every=20)
    assert n % 20 == 0 and 20 <= n < 5000 and r <= 1e-12
    dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * len(model.x_ids), model.x_ids, [0] * len(model.x_ids))
    m = dev.get_marginals(model.x_ids)
    mean = exact.grid_posterior_mean(24, 31, model.meta["y"], model.meta["r"], model.meta["qh"], model.meta["qv"])
    assert_close(m[:, 0], mean, 1e-9, "converged BP mean vs dense solve")
    n2, r2 = dev.sweep_until(1e-12, 7, check_every=3)          # max_sweeps caps the run: 3 + 3 + 1
    assert n2 in (3, 7) and r2 <= 1e-12 or n2 == 7
    with pytest.raises(cx.CortexHipError):
        dev.sweep_until(-1.0, 10)


def test_residual_of_a_numerically_broken_state_is_infinite(hip_lib):
    """a message whose precision is defined but whose mean became NaN (inf - inf, a divergent run) must not
    vanish in the max-reduction: cx_residual reports +inf, so cx_sweep_until / partition.converge never call it converged."""
    model = cx.synth.gaussian_grid(12, 12, seed=3)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.sweep(400)
    dev.residual()
    dev.sweep(2)
    assert dev.residual() < 1e-9
    # poison ONE pairwise message: natural form (xi = NaN, w = 1)
    e = int(np.flatnonzero(model.factor_kind[np.searchsorted(model.factor_ids, model.edge_fac)] == L.FACTOR_GAUSS_ADDITIVE)[0])
    dev.set_messages([model.edge_var[e]], [model.edge_fac[e]], L.TO_VARIABLE, L.FORM_NATURAL, [float("nan"), 1.0])
    assert dev.residual() == float("inf")
    n, r = dev.sweep_until(1e-9, 6, 2)
    assert n == 6 and not (r <= 1e-9)     # the NaN spreads; never "converged"


def test_residual_on_constant_zero_precision_and_beta_messages(hip_lib):
    """messages that are legitimately degenerate in MOMENT form — the Beta-Bernoulli family's natural (r, 1 - r) with a
    zero component, a Gaussian empty message (0, 0) — are constants of the model, not divergence: the residual compares snapshots
    bitwise first and the generic 2-parameter family in natural coordinates, so cx_sweep_until stops when nothing moves."""
    n = 40
    data = (np.arange(n) % 3 == 0).astype(np.float64)         # r = 1 for every third outcome: Beta(2, 1) = natural (1, 0)
    p, o, f = 1, 2 + np.arange(n), 100 + np.arange(n)
    dev = cx.DeviceGraph(schedule=L.SCHED_FLOODING, family=L.FAMILY_NATURAL2)
    dev.graph_create(np.concatenate([np.full(n, p), o]), np.concatenate([f, f]), f, np.full(n, L.FACTOR_BERNOULLI, np.int32), np.zeros(n))
    dev.set_messages(o, f, L.TO_FACTOR, L.FORM_POINT, data)
    ran, res = dev.sweep_until(1e-12, 60, 2)
    assert ran <= 6 and res == 0.0, (ran, res)
    a, b = dev.get_marginals([p])[0]
    assert (a + 1, b + 1) == (1 + data.sum(), 1 + n - data.sum())          # the exact posterior, natural parameters
    # Gaussian grid: one prior replaced by the empty (zero-precision) message (0, 0) — "no prior on this variable" — which no sweep
    # rewrites: the same value in both snapshots; its moment form is (0 * inf, inf)
    model = cx.synth.gaussian_grid(10, 10, seed=5)
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED)
    cx.synth.load_into_device(model, dev, seed_variance=1e6)
    dev.set_messages([model.prior_var[0]], [model.prior_fac[0]], L.TO_VARIABLE, L.FORM_NATURAL, [0.0, 0.0])
    ran, res = dev.sweep_until(1e-12, 3000, 10)
    assert ran < 3000 and res <= 1e-12, (ran, res)
