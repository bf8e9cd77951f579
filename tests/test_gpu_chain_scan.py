"""-m gpu: CX_SCHED_CHAIN_SCAN — one cx_sweep on a chain-structured graph equals what the reference's sequential
forward/backward schedule computes in one update_marginals! call (restated scheduler, oracle/cortex_ref.c) and the
exact Kalman smoother (oracle/exact.py).  Config C2 of BASELINE.json (T = 250,001, 1,000,002 edges) is checked at
full size against the tridiagonal solve."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from oracle import exact
from tests.helpers import assert_close, engine_oracle_from_model

pytestmark = pytest.mark.gpu


def _solve(model):
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    dev.sweep(1)
    return dev


@pytest.mark.parametrize("T,randvar", [(2, False), (3, True), (5, True), (64, True), (257, True), (1027, True), (4100, True), (70001, True)])
def test_chain_scan_equals_reference_schedule_and_exact_smoother(hip_lib, T, randvar):
    model = cx.synth.ssm_chain(T, seed=T, random_variances=randvar)
    dev = _solve(model)
    marg = dev.get_marginals(model.x_ids)
    xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
    assert_close(marg[:, 0], xm, 1e-9, "marginal mean vs tridiagonal solve")
    assert_close(marg[:, 1], xv, 1e-9, "marginal variance vs tridiagonal solve")
    if T <= 5000:
        E = engine_oracle_from_model(model)
        E.update_marginals(model.x_ids)
        _, em, ev = E.get_marginals(model.x_ids)
        assert_close(marg[:, 0], em, 1e-9, "marginal mean vs restated reference scheduler")
        assert_close(marg[:, 1], ev, 1e-9, "marginal variance vs restated reference scheduler")
        tr = model.factor_ids[T:]
        for to_variable, vs in ((True, model.x_ids[:-1]), (True, model.x_ids[1:]), (False, model.x_ids[:-1]), (False, model.x_ids[1:])):
            _, mm, mv = E.get_messages(vs, tr, to_variable=to_variable)
            got = dev.get_messages(vs, tr, L.TO_VARIABLE if to_variable else L.TO_FACTOR)
            assert_close(got[:, 0], mm, 1e-9, "chain message mean")
            assert_close(got[:, 1], mv, 1e-9, "chain message variance")
    # idempotent: a second sweep leaves the fixed point where it is
    dev.sweep(1)
    again = dev.get_marginals(model.x_ids)
    assert_close(again[:, 0], marg[:, 0], 1e-12, "second sweep mean")
    assert_close(again[:, 1], marg[:, 1], 1e-12, "second sweep variance")


@pytest.mark.parametrize("T", [3000, 70001])
def test_chain_scan_with_variances_over_twelve_decades(hip_lib, T):
    """factor variances drawn log-uniformly from [1e-6, 1e6]: the one-launch scan keeps its maps up to scale (entries scaled by powers of
    two, no division per composition) and its messages as (x, w) / d — neither may overflow or lose the small terms.  Against the
    tridiagonal solve in 50-digit arithmetic, marginal by marginal (the float64 solve is itself 2e-6 off here)."""
    model = cx.synth.ssm_chain(T, seed=T)
    rng = np.random.default_rng(T)
    r, q = 10.0 ** rng.uniform(-6, 6, T), 10.0 ** rng.uniform(-6, 6, T - 1)
    model.factor_var[:] = np.concatenate([r, q])
    dev = _solve(model)
    marg = dev.get_marginals(model.x_ids)
    xm, xv = exact.ssm_chain_posterior_decimal(model.data_y, r, q)
    assert np.all(np.isfinite(marg))
    assert_close(marg[:, 0], xm, 1e-12, "marginal mean vs tridiagonal solve")
    assert_close(marg[:, 1], xv, 1e-12, "marginal variance vs tridiagonal solve")
    assert dev.chain_scan_stats()["launches"] >= 1


def test_config_c2_full_size(hip_lib):
    """BASELINE.json configs[1]: 1M-edge scalar-Gaussian chain, one full sum-product sweep on one MI355X."""
    T = 250_001
    model = cx.synth.ssm_chain(T, seed=1234)
    assert model.n_edges == 1_000_002
    dev = _solve(model)
    assert dev.stats()["n_edges"] == 1_000_002
    marg = dev.get_marginals(model.x_ids)
    xm, xv = exact.ssm_chain_posterior(model.data_y, 1.0, 1.0)
    assert_close(marg[:, 0], xm, 1e-9, "C2 marginal mean")
    assert_close(marg[:, 1], xv, 1e-9, "C2 marginal variance")
    # the reference test's own assertions (test/inference_engine_tests.jl:485-487)
    assert np.all(marg[:, 0] >= 0) and np.all(np.diff(marg[:, 0]) >= 0) and np.all(marg[:, 1] >= 0)
    # steady state (what tools/bench_configs.py times): two launches, the scan's second kernel writes the marginals itself
    dev.profile_enable(1)
    dev.sweep(1)
    assert dev.profile_read(L.KERNEL_VAR_TO_FACTOR)[1] == 0, "the variable phase was launched on the fast path"
    dev.profile_enable(False)
    marg2 = dev.get_marginals(model.x_ids)
    assert_close(marg2[:, 0], xm, 1e-9, "C2 marginal mean, marginals written by the scan")
    assert_close(marg2[:, 1], xv, 1e-9, "C2 marginal variance, marginals written by the scan")
    # new observations go straight down the same path (no marginal off the chain depends on them)
    y2 = model.data_y + 0.5
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    xm2, xv2 = exact.ssm_chain_posterior(y2, 1.0, 1.0)
    marg3 = dev.get_marginals(model.x_ids)
    assert_close(marg3[:, 0], xm2, 1e-9, "C2 after new data, mean"); assert_close(marg3[:, 1], xv2, 1e-9, "C2 after new data, variance")


def test_several_disjoint_chains_one_scan(hip_lib):
    """segmented scan: chains of different lengths in one graph, ids interleaved."""
    lengths = [1, 2, 3, 300, 1, 1500, 7]
    ev, ef, fids, fq, dv, df, dy, xs_all, chains = [], [], [], [], [], [], [], [], []
    nid = 0
    rng = np.random.default_rng(0)
    for T in lengths:
        x = list(range(nid + 1, nid + T + 1)); y = list(range(nid + T + 1, nid + 2 * T + 1))
        lik = list(range(nid + 2 * T + 1, nid + 3 * T + 1)); tr = list(range(nid + 3 * T + 1, nid + 4 * T))
        nid += 4 * T - 1
        r, q, data = rng.uniform(0.5, 2, T), rng.uniform(0.5, 2, T - 1), rng.standard_normal(T) * 3
        for i in range(T):
            ev += [y[i], x[i]]; ef += [lik[i], lik[i]]
        for i in range(T - 1):
            ev += [x[i], x[i + 1]]; ef += [tr[i], tr[i]]
        fids += lik + tr; fq += list(r) + list(q); dv += y; df += lik; dy += list(data)
        xs_all += x; chains.append((x, data, r, q))
    model = cx.synth.Model(edge_var=np.array(ev), edge_fac=np.array(ef), factor_ids=np.array(fids),
                           factor_kind=np.full(len(fids), L.FACTOR_GAUSS_ADDITIVE, np.int32), factor_var=np.array(fq),
                           x_ids=np.array(xs_all), data_var=np.array(dv), data_fac=np.array(df), data_y=np.array(dy))
    dev = _solve(model)
    for x, data, r, q in chains:
        marg = dev.get_marginals(x)
        if len(x) == 1:   # a lone variable: marginal = its likelihood message
            assert_close(marg[:, 0], data, 1e-12, "lone variable mean")
            assert_close(marg[:, 1], r, 1e-12, "lone variable variance")
            continue
        xm, xv = exact.ssm_chain_posterior(data, r, q)
        assert_close(marg[:, 0], xm, 1e-9, f"chain of {len(x)} mean")
        assert_close(marg[:, 1], xv, 1e-9, f"chain of {len(x)} variance")


def test_marginals_written_by_the_scan_on_segmented_chains_and_messages_on_demand(hip_lib):
    """several paths in one graph (no lone variables, so every reader of messages is on a chain): from the second sweep on the
    scan's apply kernel writes the marginals — seams between link blocks, heads and tails of paths included — and
    variable→factor messages are recomputed when asked for; both equal the first sweep's (variable phase) to rounding."""
    lengths = [2, 1500, 3, 1023, 1025, 2049, 7]
    ev, ef, fids, fq, dv, df, dy, xs_all, trs = [], [], [], [], [], [], [], [], []
    nid = 0
    rng = np.random.default_rng(1)
    for T in lengths:
        x = list(range(nid + 1, nid + T + 1)); y = list(range(nid + T + 1, nid + 2 * T + 1))
        lik = list(range(nid + 2 * T + 1, nid + 3 * T + 1)); tr = list(range(nid + 3 * T + 1, nid + 4 * T))
        nid += 4 * T - 1
        for i in range(T):
            ev += [y[i], x[i]]; ef += [lik[i], lik[i]]
        for i in range(T - 1):
            ev += [x[i], x[i + 1]]; ef += [tr[i], tr[i]]
            trs.append((x[i], x[i + 1], tr[i]))
        fids += lik + tr; fq += list(rng.uniform(0.5, 2, T)) + list(rng.uniform(0.5, 2, T - 1)); dv += y; df += lik; dy += list(rng.standard_normal(T) * 3)
        xs_all += x
    model = cx.synth.Model(edge_var=np.array(ev), edge_fac=np.array(ef), factor_ids=np.array(fids),
                           factor_kind=np.full(len(fids), L.FACTOR_GAUSS_ADDITIVE, np.int32), factor_var=np.array(fq),
                           x_ids=np.array(xs_all), data_var=np.array(dv), data_fac=np.array(df), data_y=np.array(dy))
    dev = _solve(model)                                   # first sweep: full variable phase
    first = dev.get_marginals(model.x_ids)
    left = np.array([a for a, _, _ in trs]); right = np.array([b for _, b, _ in trs]); tf = np.array([t for _, _, t in trs])
    v2f_first = [dev.get_messages(left, tf, L.TO_FACTOR), dev.get_messages(right, tf, L.TO_FACTOR)]
    dev.profile_enable(1)
    dev.sweep(1)
    assert dev.profile_read(L.KERNEL_VAR_TO_FACTOR)[1] == 0
    dev.profile_enable(False)
    second = dev.get_marginals(model.x_ids)
    assert_close(second[:, 0], first[:, 0], 1e-12, "mean"); assert_close(second[:, 1], first[:, 1], 1e-12, "variance")
    for a, b in zip(v2f_first, [dev.get_messages(left, tf, L.TO_FACTOR), dev.get_messages(right, tf, L.TO_FACTOR)]):
        assert_close(b[:, 0], a[:, 0], 1e-12, "variable→factor mean on demand"); assert_close(b[:, 1], a[:, 1], 1e-12, "variable→factor variance on demand")


@pytest.mark.parametrize("n,extra", [(7, 0), (1500, 2), (2300, 5)])
def test_chain_scan_with_linear_factors_and_several_side_factors(hip_lib, n, extra):
    """x_{t+1} = a_t x_t + b_t + N(0, q_t) with per-factor parameters, and `extra` more observations per state (side sums of
    several messages): the exact posterior from the joint precision matrix, on the sweep that runs the variable phase and on
    the following ones (marginals written by the scan), and again after new data."""
    rng = np.random.default_rng(n + extra)
    m = 1 + extra                                           # observations per state
    x = np.arange(1, n + 1)
    yv = (n + 1 + np.arange(n * m)).reshape(n, m); lk = (n + n * m + 1 + np.arange(n * m)).reshape(n, m)
    tr = n + 2 * n * m + 1 + np.arange(n - 1)
    ev = np.concatenate([yv.ravel(), np.repeat(x, m), x[:-1], x[1:]]); ef = np.concatenate([lk.ravel(), lk.ravel(), tr, tr])
    role = np.concatenate([np.full(n * m, L.ROLE_OUT), np.full(n * m, L.ROLE_IN), np.full(n - 1, L.ROLE_IN), np.full(n - 1, L.ROLE_OUT)]).astype(np.int32)
    fids = np.concatenate([lk.ravel(), tr])
    kinds = np.concatenate([np.full(n * m, L.FACTOR_GAUSS_ADDITIVE), np.full(n - 1, L.FACTOR_GAUSS_LINEAR)]).astype(np.int32)
    r = rng.uniform(0.5, 2.0, (n, m)); q = rng.uniform(0.3, 1.5, n - 1); a = rng.uniform(0.6, 1.1, n - 1); b = rng.uniform(-0.5, 0.5, n - 1)
    params = np.zeros((n * m + n - 1, L.NPARAM)); params[:n * m, 0] = r.ravel(); params[n * m:, 0] = q; params[n * m:, 1] = a; params[n * m:, 2] = b

    def exact_posterior(ys):
        from scipy.linalg import solveh_banded
        d0 = (1.0 / r).sum(axis=1); h = (ys / r).sum(axis=1); d1 = np.zeros(n - 1)
        d0[:-1] += a * a / q; d0[1:] += 1.0 / q; d1 -= a / q
        h[:-1] -= a * b / q; h[1:] += b / q
        ab = np.zeros((2, n)); ab[1] = d0; ab[0, 1:] = d1
        mean = solveh_banded(ab, h)
        # diagonal of the inverse of a tridiagonal matrix by the two-sided recursion
        fw = np.zeros(n); bw = np.zeros(n)
        fw[0] = d0[0]
        for i in range(1, n):
            fw[i] = d0[i] - d1[i - 1] ** 2 / fw[i - 1]
        bw[-1] = d0[-1]
        for i in range(n - 2, -1, -1):
            bw[i] = d0[i] - d1[i] ** 2 / bw[i + 1]
        return mean, 1.0 / (fw + bw - d0)

    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    dev.graph_create(ev, ef, fids, kinds, params, edge_role=role)
    ys = rng.standard_normal((n, m)) * 2
    dev.set_messages(yv.ravel(), lk.ravel(), L.TO_FACTOR, L.FORM_POINT, ys.ravel())
    em, evar = exact_posterior(ys)
    for sweep in range(3):
        dev.sweep(1)
        marg = dev.get_marginals(x)
        assert_close(marg[:, 0], em, 1e-9, f"sweep {sweep}: mean"); assert_close(marg[:, 1], evar, 1e-9, f"sweep {sweep}: variance")
    ys2 = ys + rng.standard_normal((n, m))
    dev.set_messages(yv.ravel(), lk.ravel(), L.TO_FACTOR, L.FORM_POINT, ys2.ravel())
    dev.sweep(1)
    em2, _ = exact_posterior(ys2)
    marg = dev.get_marginals(x)
    assert_close(marg[:, 0], em2, 1e-9, "after new data: mean"); assert_close(marg[:, 1], evar, 1e-9, "after new data: variance")


def test_prescanned_tile_totals_path(hip_lib):
    """chains of more than 8192 tiles (8.4 M links) pre-scan the tile totals with a one-workgroup kernel instead of composing the
    carry in every apply workgroup; CX_CHAIN_OWN_CARRY_TILES lowers that threshold so the path runs at T = 70,001 (69 tiles) —
    in a child process, the library reads the variable once"""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import numpy as np
        import cortex.jl_amd as cx
        from cortex.jl_amd import _lib as L
        from oracle import exact
        T = 70001
        model = cx.synth.ssm_chain(T, seed=11, random_variances=True)
        dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
        cx.synth.load_into_device(model, dev)
        xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
        for sweep in range(3):          # variable phase first, then the marginals the scan writes
            dev.sweep(1)
            m = dev.get_marginals(model.x_ids)
            assert np.max(np.abs(m[:, 0] - xm) / (1e-300 + np.abs(xm).max())) < 1e-9, sweep
            assert np.max(np.abs(m[:, 1] - xv) / xv) < 1e-9, sweep
        print("ok")
    """)
    env = dict(os.environ, CX_CHAIN_OWN_CARRY_TILES="8")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_chain_scan_refuses_loopy_graphs(hip_lib):
    model = cx.synth.gaussian_grid(4, 4, seed=1)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev, seed_variance=10.0)
    with pytest.raises(cx.CortexHipError) as e:
        dev.sweep(1)
    assert e.value.code == L.ERR_UNSUPPORTED and "chain" in e.value.message


def test_chain_longer_than_one_chunk_of_tile_totals(hip_lib):
    """T = 1,100,000: 1075 tiles per direction, i.e. the apply kernel composes its carry from TWO chunks of tile totals."""
    T = 1_100_000
    model = cx.synth.ssm_chain(T, seed=5)
    dev = _solve(model)
    idx = np.concatenate([np.arange(0, 1500), np.arange(T // 2 - 750, T // 2 + 750), np.arange(T - 1500, T)])
    marg = dev.get_marginals(model.x_ids[idx])
    xm, xv = exact.ssm_chain_posterior(model.data_y, 1.0, 1.0)
    assert_close(marg[:, 0], xm[idx], 1e-9, "long chain marginal mean")
    assert_close(marg[:, 1], xv[idx], 1e-9, "long chain marginal variance")


def test_chain_scan_sees_data_changes_between_sweeps(hip_lib):
    """The side sums / leaf messages are cached between sweeps: new observations (cx_set_messages) must invalidate them."""
    T = 300
    model = cx.synth.ssm_chain(T, seed=3)
    dev = _solve(model)
    first = dev.get_marginals(model.x_ids)
    dev.sweep(1)                                                      # cached side sums: same answer
    second = dev.get_marginals(model.x_ids)
    # (from the second sweep on the scan's own kernel writes the chain variables' marginals, as side + beta + alpha instead of the
    # variable phase's slot order: equal to rounding, and bit for bit from then on)
    assert_close(second[:, 0], first[:, 0], 1e-13, "second sweep mean"); assert_close(second[:, 1], first[:, 1], 1e-13, "second sweep variance")
    dev.sweep(1)
    assert np.array_equal(dev.get_marginals(model.x_ids), second)
    y2 = model.data_y + np.linspace(-3, 3, T)
    dev.set_messages(model.data_var, model.data_fac, L.TO_FACTOR, L.FORM_POINT, y2)
    dev.sweep(1)
    xm, xv = exact.ssm_chain_posterior(y2, 1.0, 1.0)
    marg = dev.get_marginals(model.x_ids)
    assert_close(marg[:, 0], xm, 1e-9, "posterior mean after new data")
    assert_close(marg[:, 1], xv, 1e-9, "posterior variance after new data")
    assert not np.allclose(marg[:, 0], first[:, 0])


@pytest.mark.parametrize("T", [1500, 70001, 250001, 400001])
def test_the_one_launch_scan_equals_the_two_launch_scan(hip_lib, monkeypatch, T):
    """(round 6) the chain scan as ONE launch (tile totals published with the launch's tag, csrc/cx_chain.hip: k_chain_onepass) against
    the two launches it replaces (CX_CHAIN_ONEPASS=0, read when a handle first scans): the same maps composed in the same order, the one
    launch without divisions (entries scaled by powers of two) — equal to a few units in the last place —, sweep after sweep (the epoch
    moves on inside the launch).  T = 400,001 is 391 tiles: more than one workgroup per compute unit, the instance held to 128 registers."""
    model = cx.synth.ssm_chain(T, seed=T + 1, random_variances=True)
    a = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, a)
    monkeypatch.setenv("CX_CHAIN_ONEPASS", "0")
    b = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, b)
    b.sweep(1)
    monkeypatch.delenv("CX_CHAIN_ONEPASS")
    tr = model.factor_ids[T:]
    for sweep in range(4):
        a.sweep(1)
        if sweep:
            b.sweep(1)
        np.testing.assert_allclose(a.get_marginals(model.x_ids), b.get_marginals(model.x_ids), rtol=1e-13, atol=0, err_msg=f"sweep {sweep}")
        for vs in (model.x_ids[:-1], model.x_ids[1:]):
            np.testing.assert_allclose(a.get_messages(vs, tr, L.TO_VARIABLE, L.FORM_NATURAL), b.get_messages(vs, tr, L.TO_VARIABLE, L.FORM_NATURAL), rtol=1e-13, atol=1e-300)
    sa, sb = a.chain_scan_stats(), b.chain_scan_stats()
    assert sa["state"] == 1 and sa["launches"] == 4, sa
    assert sb["state"] == -1 and sb["launches"] == 0, sb


def test_a_one_launch_scan_whose_wait_times_out_fails_loudly_and_the_sweep_can_be_repeated(hip_lib, monkeypatch):
    """fault injection: the first tile never publishes its forward total (what a workgroup that never becomes resident looks like).  The
    waits are bounded in time; nobody stores anything; the next call that looks at the device returns CX_ERR_DEVICE; the handle goes
    back to two launches, and repeating the sweep gives the exact result."""
    T = 9000
    model = cx.synth.ssm_chain(T, seed=5, random_variances=True)
    dev = cx.DeviceGraph(schedule=L.SCHED_CHAIN_SCAN)
    cx.synth.load_into_device(model, dev)
    monkeypatch.setenv("CX_CHAIN_ONEPASS_FAULT", "1")
    monkeypatch.setenv("CX_CHAIN_ONEPASS_TIMEOUT_MS", "20")
    with pytest.raises(cx.CortexHipError, match="one-launch chain scan timed out") as ei:
        dev.sweep(1)
        dev.sync()
        dev.get_marginals(model.x_ids)
    assert ei.value.code == L.ERR_DEVICE
    monkeypatch.delenv("CX_CHAIN_ONEPASS_FAULT")
    assert dev.chain_scan_stats()["state"] == -1
    dev.sweep(1)
    marg = dev.get_marginals(model.x_ids)
    xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
    assert_close(marg[:, 0], xm, 1e-9, "marginal mean after the repeated sweep")
    assert_close(marg[:, 1], xv, 1e-9, "marginal variance after the repeated sweep")
    assert dev.chain_scan_stats()["launches"] == 1
