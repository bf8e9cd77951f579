"""-m gpu: two sweeps per launch (cx_tiles.hip, cx_config.sweeps_per_launch = 2; an opt-in experiment) must be the SAME computation as two single-sweep
launches — every factor→variable message, every marginal and the UndefValue() pattern bit for bit — on grids, chains with
observed variables, linear factors, random loopy graphs and deep-halo partitions, for even and odd sweep counts, and around
everything that touches the retained buffer (variable→factor read-back, checkpoints, data injection between sweeps)."""
import threading

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import _lib as L
from cortex.jl_amd import partition
from tests.helpers import random_loopy_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_tiled_build(hip_lib):
    """cx_tiles.hip is built on request only since round 4 (CX_BUILD_TILED2=1 python -m cortex.jl_amd.build --force: measured slower than
    single sweeps, kept as source and test): without it sweeps_per_launch = 2 runs single sweeps and these tests have nothing to compare"""
    probe = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=2)
    cx.synth.load_into_device(cx.synth.gaussian_grid(20, 20, seed=1), probe, 1e6)
    probe.sweep(2)
    built = probe.tile_stats()["n_tiles"] > 0
    probe.close()
    if not built:
        pytest.skip("libcortex_hip.so was built without cx_tiles.hip (CX_BUILD_TILED2)")


def _pair(model, seed_variance=None, **kw):
    one = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=1, **kw)
    two = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=2, **kw)
    for d in (one, two):
        cx.synth.load_into_device(model, d, seed_variance)
    return one, two


def _same_state(a, b, model, what):
    ev, ef = model.edge_var, model.edge_fac
    x = a.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    y = b.get_messages(ev, ef, L.TO_VARIABLE, L.FORM_NATURAL)
    assert np.array_equal(x, y, equal_nan=True), f"{what}: factor→variable messages differ"
    ids = np.unique(ev)
    assert np.array_equal(a.get_marginals(ids), b.get_marginals(ids), equal_nan=True), f"{what}: marginals differ"
    x = a.get_messages(ev, ef, L.TO_FACTOR, L.FORM_NATURAL)     # regenerates time n - 1 on the tiled handle
    y = b.get_messages(ev, ef, L.TO_FACTOR, L.FORM_NATURAL)
    assert np.array_equal(x, y, equal_nan=True), f"{what}: variable→factor messages differ"


@pytest.mark.parametrize("shape", [(37, 23), (64, 64), (5, 300), (130, 97)])
def test_two_sweeps_per_launch_equal_single_sweeps_on_grids(hip_lib, shape):
    model = cx.synth.gaussian_grid(*shape, seed=7)
    one, two = _pair(model, seed_variance=1e6)
    done = 0
    for n in (2, 1, 4, 3, 6, 7):         # even counts: pairs only; odd: pairs + one plain sweep; a lone sweep: plain
        one.sweep(n); two.sweep(n)
        done += n
        _same_state(one, two, model, f"{shape} after {done} sweeps")
    st = two.tile_stats()
    assert st["n_tiles"] >= 1 and 1.0 < st["variables_loaded_per_owned"] < 4.0 and st["lds_bytes_per_workgroup"] <= 160 * 1024
    assert one.tile_stats()["n_tiles"] == 0
    assert one.stats()["sweeps_done"] == two.stats()["sweeps_done"] == done


@pytest.mark.parametrize("T,randvar", [(3, False), (40, True), (1000, True)])
def test_two_sweeps_per_launch_on_chains_with_observed_variables(hip_lib, T, randvar):
    """state-space chain, data on degree-1 observed variables, NO seeds: the wavefront of defined messages advances one step
    per sweep, so every intermediate state carries UndefValue()s that both paths must agree on"""
    model = cx.synth.ssm_chain(T, seed=5, random_variances=randvar)
    one, two = _pair(model)
    for n in (2, 2, 3, 4, 5):
        one.sweep(n); two.sweep(n)
        _same_state(one, two, model, f"chain T={T}")
    one.sweep(T + 3); two.sweep(T + 3)
    _same_state(one, two, model, f"chain T={T}, converged")


def test_two_sweeps_per_launch_with_linear_factors_and_fresh_data(hip_lib):
    """x_{t+1} = a x_t + b + N(0, q) factors (the LINEAR instantiation); new data injected between sweeps"""
    n, a, b, q, r = 60, 0.9, 0.3, 0.5, 0.7
    x = np.arange(1, n + 1); y = x + n; lik = x + 2 * n; tr = np.arange(3 * n + 1, 4 * n)
    ev = np.concatenate([y, x, x[:-1], x[1:]]); ef = np.concatenate([lik, lik, tr, tr])
    role = np.concatenate([np.full(n, L.ROLE_OUT), np.full(n, L.ROLE_IN), np.full(n - 1, L.ROLE_IN), np.full(n - 1, L.ROLE_OUT)]).astype(np.int32)
    fids = np.concatenate([lik, tr])
    kinds = np.concatenate([np.full(n, L.FACTOR_GAUSS_ADDITIVE), np.full(n - 1, L.FACTOR_GAUSS_LINEAR)]).astype(np.int32)
    params = np.zeros((2 * n - 1, L.NPARAM)); params[:n, 0] = r; params[n:, 0] = q; params[n:, 1] = a; params[n:, 2] = b
    rng = np.random.default_rng(2)
    data = rng.standard_normal(n)
    devs = []
    for spl in (1, 2):
        d = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=spl)
        d.graph_create(ev, ef, fids, kinds, params, edge_role=role)
        d.set_messages(y, lik, L.TO_FACTOR, L.FORM_POINT, data)
        devs.append(d)
    one, two = devs

    class M:
        edge_var, edge_fac = ev, ef
    for k in (4, 3, 8):
        one.sweep(k); two.sweep(k)
        _same_state(one, two, M, "linear chain")
    fresh = rng.standard_normal(n)
    for d in (one, two):
        d.set_messages(y, lik, L.TO_FACTOR, L.FORM_POINT, fresh)
        d.set_messages(x[:3], lik[:3], L.TO_VARIABLE, L.FORM_MOMENT, np.tile([0.5, 2.0], 3))    # user-set f2v: written to both buffers
    for k in (2, 5, 2 * n):
        one.sweep(k); two.sweep(k)
        _same_state(one, two, M, "linear chain after new data")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_two_sweeps_per_launch_on_random_loopy_graphs(hip_lib, seed):
    model, _owner = random_loopy_model(seed, 2, nv=900, extra=500)
    one, two = _pair(model, seed_variance=1e6)
    for n in (2, 3, 6):
        one.sweep(n); two.sweep(n)
        _same_state(one, two, model, f"random graph {seed}")


def test_checkpoint_and_residual_around_two_sweep_launches(hip_lib):
    model = cx.synth.gaussian_grid(40, 33, seed=4)
    one, two = _pair(model, seed_variance=1e6)
    one.sweep(6); two.sweep(6)
    blob = two.export_state()                       # normalises the retained buffer to time n - 1
    assert np.array_equal(one.export_state(), blob)  # the same blob as the single-sweep handle's, byte for byte
    _same_state(one, two, model, "after export")
    two.sweep(5); one.sweep(5)
    third = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=2)
    cx.synth.load_into_device(model, third, 3.0)
    third.import_state(blob)
    third.sweep(5)
    _same_state(one, third, model, "restored handle")
    one.residual(); two.residual()
    one.sweep(4); two.sweep(4)
    assert one.residual() == two.residual()
    n1, r1 = one.sweep_until(1e-12, 3000, 10)
    n2, r2 = two.sweep_until(1e-12, 3000, 10)
    assert (n1, r1) == (n2, r2)


def test_graphs_the_tiles_cannot_take_fall_back_to_single_sweeps(hip_lib):
    """a hub of degree 40 (CSR tail, wave-scan kernels): cx_sweep(n >= 2) keeps working, one launch per sweep"""
    n = 40
    ev = np.concatenate([np.full(n, 1), 2 + np.arange(n)]); ef = np.concatenate([100 + np.arange(n), 100 + np.arange(n)])
    dev = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=2)
    dev.graph_create(ev, ef, 100 + np.arange(n), np.full(n, L.FACTOR_GAUSS_ADDITIVE, np.int32), np.ones(n))
    dev.set_messages(2 + np.arange(n), 100 + np.arange(n), L.TO_FACTOR, L.FORM_POINT, np.arange(n, dtype=float))
    dev.sweep(4)
    assert dev.tile_stats()["n_tiles"] == 0
    m = dev.get_marginals([1])[0]
    assert m[0] == pytest.approx(np.arange(n).mean(), rel=1e-12) and m[1] == pytest.approx(1.0 / n, rel=1e-12)


@pytest.mark.parametrize("world,rows,cols,depth", [(3, 40, 70, 4), (2, 30, 50, 3)])
def test_deep_halo_partition_with_two_sweep_launches(hip_lib, world, rows, cols, depth):
    """strips with a deep halo whose handles run the two-sweep launches between exchanges (depth even and odd): owned
    messages and marginals equal the un-partitioned single-sweep device run bit for bit"""
    import torch

    from tests.test_gpu_partition import LoopbackDist

    sweeps = 3 * depth + 1
    whole_model = cx.synth.gaussian_grid(rows * world, cols, seed=21)
    whole = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=1)
    cx.synth.load_into_device(whole_model, whole, seed_variance=1e6)
    whole.sweep(sweeps)
    ld = LoopbackDist(world, torch)
    devs, parts, errors = [None] * world, [None] * world, []

    def run(rank):
        try:
            ld.bind(rank)
            part = partition.grid_strip_deep(rows, cols, rank, world, depth, seed=21)
            dev = cx.DeviceGraph(schedule=L.SCHED_FUSED, sweeps_per_launch=2)
            cx.synth.load_into_device(part.model, dev, seed_variance=1e6)
            ex = partition.DeepHaloExchange(partition.DeviceStateSweeper(dev, part, torch, torch.device("cuda", 0)), part, ld)
            ex.sweep(sweeps)
            dev.sync()
            assert dev.tile_stats()["n_tiles"] > 0
            devs[rank], parts[rank] = dev, part
        except Exception as e:  # pragma: no cover
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for rank in range(world):
        ids = parts[rank].owned_x
        assert np.array_equal(devs[rank].get_marginals(ids), whole.get_marginals(ids), equal_nan=True), f"rank {rank}"
        m = parts[rank].model
        own = np.isin(m.edge_var, ids)
        assert np.array_equal(devs[rank].get_messages(m.edge_var[own], m.edge_fac[own], L.TO_VARIABLE, L.FORM_NATURAL),
                              whole.get_messages(m.edge_var[own], m.edge_fac[own], L.TO_VARIABLE, L.FORM_NATURAL), equal_nan=True)


def test_config_c4_full_size_two_sweeps_per_launch(hip_lib):
    """BASELINE config C4 at full size (1415 x 1415, 10,005,465 edges): 20 sweeps as ten two-sweep launches equal 20 single
    sweeps bit for bit — every message and every marginal"""
    model = cx.synth.gaussian_grid(1415, 1415, seed=1234)
    one, two = _pair(model, seed_variance=1e6)
    one.sweep(20); two.sweep(20)
    assert two.tile_stats()["n_tiles"] > 1000
    sel = np.arange(0, model.n_edges, 7)
    a = one.get_messages(model.edge_var[sel], model.edge_fac[sel], L.TO_VARIABLE, L.FORM_NATURAL)
    b = two.get_messages(model.edge_var[sel], model.edge_fac[sel], L.TO_VARIABLE, L.FORM_NATURAL)
    assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(one.get_marginals(model.x_ids), two.get_marginals(model.x_ids), equal_nan=True)
