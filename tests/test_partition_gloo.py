"""The N > 1 path on CPU: world_size-2 and -3 gloo jobs running the product's partitioner and halo exchange
(cortex.jl_amd/partition.py) around the CPU checker must reproduce the single-process sweep of the whole grid
bit for bit (same arithmetic, same order; only the cut messages travel)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import partition
from tests.helpers import flood_oracle_from_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_strip_partition_covers_the_grid_exactly_once():
    rows, cols, world = 5, 6, 3
    whole = cx.synth.gaussian_grid(rows * world, cols, seed=99)
    owned, factors = [], {}
    for r in range(world):
        p = partition.grid_strip(rows, cols, r, world, seed=99)
        owned.append(p.model.x_ids)
        for f, k, q in zip(p.model.factor_ids, p.model.factor_kind, p.model.factor_var):
            assert factors.setdefault(int(f), (int(k), float(q))) == (int(k), float(q))  # cut factors agree on both sides
        # every exported message is imported by the peer as the same (variable, factor) pair
        for peer in p.peers:
            q = partition.grid_strip(rows, cols, peer.rank, world, seed=99)
            back = [pp for pp in q.peers if pp.rank == r][0]
            assert np.array_equal(p.send_var[peer.send], q.recv_var[back.recv])
            assert np.array_equal(p.send_fac[peer.send], q.recv_fac[back.recv])
    assert np.array_equal(np.sort(np.concatenate(owned)), whole.x_ids)
    assert sorted(factors) == sorted(int(f) for f in whole.factor_ids)
    wf = dict(zip(whole.factor_ids.tolist(), whole.factor_var.tolist()))
    assert all(wf[f] == q for f, (_k, q) in factors.items())


@pytest.mark.parametrize("world,depth", [(2, 0), (3, 0), (2, 1), (3, 2), (3, 3), (2, 4)])
def test_gloo_halo_exchange_matches_single_process(tmp_path, world, depth):
    """depth 0: one message halo per sweep; depth w: deep halo, the redundant rows' state exchanged every w sweeps.  Either
    way every message and marginal of an owned variable equals the single-process sweep bit for bit."""
    rows, cols, sweeps = 4, 7, 9
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(rows), str(cols),
                                       str(sweeps), out] + ([str(depth)] if depth else []), env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    whole = cx.synth.gaussian_grid(rows * world, cols, seed=99)
    g = flood_oracle_from_model(whole, 1e6)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    key = lambda v, f: v.astype(np.int64) * (1 << 32) + f  # noqa: E731
    gkey = key(g.edge_var, g.edge_fac)
    order = np.argsort(gkey)
    seen = 0
    for r in range(world):
        d = np.load(out + f".rank{r}.npz")
        assert bool(d["audit_ok"]) and not bool(d["audit_bad"])     # partition.verify_last_exchange, as bench.py runs it
        own = np.isin(d["edge_var"], d["owned"])
        pos = order[np.searchsorted(gkey[order], key(d["edge_var"][own], d["edge_fac"][own]))]
        for name in ("f2v_m", "f2v_v", "v2f_m", "v2f_v"):
            a, b = d[name][own], getattr(g, name)[pos]
            assert np.array_equal(np.isnan(a), np.isnan(b)), name
            assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), f"{name} differs on rank {r}"  # bit-exact
        vi = np.searchsorted(g.var_ids, d["owned"])
        li = np.searchsorted(d["var_ids"], d["owned"])
        assert np.array_equal(d["marg_m"][li], gm[vi]) and np.array_equal(d["marg_v"][li], gv[vi])
        seen += own.sum()
    assert seen == g.ne


def _union_check(whole, parts):
    owned = np.concatenate([p.model.x_ids for p in parts])
    assert np.array_equal(np.sort(owned), np.sort(whole.x_ids))
    for p in parts:
        for peer in p.peers:
            q = parts[peer.rank]
            back = [pp for pp in q.peers if pp.rank == p.rank][0]
            assert np.array_equal(p.send_var[peer.send], q.recv_var[back.recv])
            assert np.array_equal(p.send_fac[peer.send], q.recv_fac[back.recv])
            assert np.array_equal(q.send_var[back.send], p.recv_var[peer.recv])


def test_generic_partitioner_reproduces_the_strip_partition():
    rows, cols, world = 4, 6, 3
    whole = cx.synth.gaussian_grid(rows * world, cols, seed=99)
    parts = [partition.contiguous_blocks(whole, r, world) for r in range(world)]
    _union_check(whole, parts)
    for r in range(world):
        s = partition.grid_strip(rows, cols, r, world, seed=99)
        g = parts[r]
        key = lambda m: sorted(zip(m.edge_var.tolist(), m.edge_fac.tolist()))  # noqa: E731
        assert key(g.model) == key(s.model)
        assert sorted(zip(g.send_var.tolist(), g.send_fac.tolist())) == sorted(zip(s.send_var.tolist(), s.send_fac.tolist()))


def test_generic_partitioner_on_a_chain_matches_single_process():
    """time blocks of a state-space chain: 1 cut factor per boundary; flooding sweeps with the exchange equal the
    un-partitioned sweeps bit for bit (in-process exchange through numpy, same code path as _dist_worker)."""
    from tests._dist_worker import OracleSweeper

    T, world, sweeps = 40, 4, 30
    whole = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    parts = [partition.contiguous_blocks(whole, r, world) for r in range(world)]
    _union_check(whole, parts)
    assert all(len(p.send_var) == (1 if r in (0, world - 1) else 2) for r, p in enumerate(parts))
    sws = [OracleSweeper(p, None) for p in parts]
    for _ in range(sweeps):
        for sw in sws:
            sw.sweep_begin()
        for r, p in enumerate(parts):
            for peer in p.peers:
                back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
        for sw in sws:
            sw.sweep_main(); sw.sweep_end()
    g = flood_oracle_from_model(whole)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    for p, sw in zip(parts, sws):
        m, v = sw.g.marginals()
        li = np.searchsorted(sw.g.var_ids, p.model.x_ids); wi = np.searchsorted(g.var_ids, p.model.x_ids)
        assert np.array_equal(m[li], gm[wi], equal_nan=True) and np.array_equal(v[li], gv[wi], equal_nan=True)


def test_generic_deep_partitioner_reproduces_the_deep_strips():
    rows, cols, world, depth = 5, 6, 3, 2
    whole = cx.synth.gaussian_grid(rows * world, cols, seed=99)
    for r in range(world):
        s = partition.grid_strip_deep(rows, cols, r, world, depth, seed=99)
        g = partition.contiguous_blocks(whole, r, world, depth=depth)
        key = lambda m: sorted(zip(m.edge_var.tolist(), m.edge_fac.tolist()))  # noqa: E731
        assert key(g.model) == key(s.model)
        for name in ("send_var", "send_fac", "recv_var", "recv_fac"):
            assert np.array_equal(getattr(g, name), getattr(s, name)), name
        assert [p.rank for p in g.peers] == [p.rank for p in s.peers]
        assert np.array_equal(np.sort(g.owned_x), np.sort(s.owned_x))


@pytest.mark.parametrize("depth", [1, 3])
def test_generic_deep_partition_of_a_chain_matches_single_process(depth):
    """time blocks of a state-space chain with a deep halo (observations are variables of their own here): flooding sweeps
    with one state exchange per `depth` sweeps equal the un-partitioned sweeps bit for bit on every owned variable."""
    from tests._dist_worker import OracleStateSweeper

    T, world, sweeps = 40, 4, 2 * depth + 5
    whole = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    parts = [partition.contiguous_blocks(whole, r, world, depth=depth) for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate([p.owned_x for p in parts])), np.sort(whole.x_ids))
    sws = [OracleStateSweeper(p, None) for p in parts]
    for k in range(sweeps):
        if k % depth == 0:
            for sw in sws:
                sw.pack()
            for r, p in enumerate(parts):
                for peer in p.peers:
                    back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                    assert back.recv.stop - back.recv.start == peer.send.stop - peer.send.start
                    sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
            for sw in sws:
                sw.unpack()
        for sw in sws:
            sw.sweep()
    g = flood_oracle_from_model(whole)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    for p, sw in zip(parts, sws):
        m, v = sw.g.marginals()
        li = np.searchsorted(sw.g.var_ids, p.owned_x); wi = np.searchsorted(g.var_ids, p.owned_x)
        assert np.array_equal(m[li], gm[wi], equal_nan=True) and np.array_equal(v[li], gv[wi], equal_nan=True)
        own = np.isin(sw.g.edge_var, p.owned_x)
        e = g.edge_index(sw.g.edge_var[own], sw.g.edge_fac[own])
        for name in ("f2v_m", "f2v_v", "v2f_m", "v2f_v"):
            assert np.array_equal(getattr(sw.g, name)[own], getattr(g, name)[e], equal_nan=True), name


def test_auto_partition_falls_back_to_blocks_without_metis():
    """libmetis is not in this image: metis_assignment reports that, auto_partition cuts contiguous blocks."""
    whole = cx.synth.gaussian_grid(8, 5, seed=3)
    if partition.metis_assignment(whole, 2) is None:
        a, b = partition.auto_partition(whole, 1, 2, depth=2), partition.contiguous_blocks(whole, 1, 2, depth=2)
        assert np.array_equal(a.send_var, b.send_var) and np.array_equal(a.recv_fac, b.recv_fac)
    else:   # a box with METIS: the cut must still cover every latent variable exactly once
        parts = [partition.auto_partition(whole, r, 2, depth=1) for r in range(2)]
        assert np.array_equal(np.sort(np.concatenate([p.owned_x for p in parts])), np.sort(whole.x_ids))


@pytest.mark.parametrize("seed,world,depth", [(0, 2, 1), (1, 3, 2), (2, 4, 3), (3, 3, 1)])
def test_generic_deep_partition_of_random_sparse_graphs(seed, world, depth):
    """Random loopy models (unary priors + random pairwise factors), random variable→rank maps, deep halo of depth 1–3:
    in-process exchange with the CPU checker as the sweeper, every owned message and marginal bit-identical to the
    single-process flooding sweeps."""
    from tests._dist_worker import OracleStateSweeper

    from tests.helpers import random_loopy_model

    whole, owner = random_loopy_model(seed, world)
    x = whole.x_ids
    parts = [partition.by_assignment_deep(whole, owner, r, world, depth) for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate([p.owned_x for p in parts])), x)
    sws = [OracleStateSweeper(p, 1e6) for p in parts]
    sweeps = 3 * depth + 2
    for k in range(sweeps):
        if k % depth == 0:
            for sw in sws:
                sw.pack()
            for r, p in enumerate(parts):
                for peer in p.peers:
                    back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                    assert back.recv.stop - back.recv.start == peer.send.stop - peer.send.start
                    sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
            for sw in sws:
                sw.unpack()
        for sw in sws:
            sw.sweep()
    g = flood_oracle_from_model(whole, 1e6)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    for p, sw in zip(parts, sws):
        if len(p.owned_x) == 0:
            continue
        m, v = sw.g.marginals()
        li = np.searchsorted(sw.g.var_ids, p.owned_x); wi = np.searchsorted(g.var_ids, p.owned_x)
        assert np.array_equal(m[li], gm[wi], equal_nan=True) and np.array_equal(v[li], gv[wi], equal_nan=True)
        own = np.isin(sw.g.edge_var, p.owned_x)
        e = g.edge_index(sw.g.edge_var[own], sw.g.edge_fac[own])
        for name in ("f2v_m", "f2v_v"):
            assert np.array_equal(getattr(sw.g, name)[own], getattr(g, name)[e], equal_nan=True), name


@pytest.mark.parametrize("seed,world,depth,n_factors", [(0, 2, 1, 40), (1, 3, 2, 120), (2, 4, 1, 500)])
def test_deep_partition_of_graphs_with_factors_of_more_than_two_variables(seed, world, depth, n_factors):
    """(round 5) loopy models of linear-Gaussian factors of 3 - 7 variables cut with a deep halo: a cut factor keeps all its variables
    on every rank that holds one of them within the halo.  In-process exchange, the k-ary CPU checker as the sweeper: owned messages
    and marginals bit-identical to the single-process sweeps."""
    from tests._dist_worker import OracleStateSweeper

    whole = cx.synth.kary_model(n_factors, seed=60 + seed, tree=False)
    used = np.unique(whole.edge_var)
    owner_map = np.random.default_rng(seed).integers(0, world, int(used.max()) + 1)
    owner = lambda ids: owner_map[np.asarray(ids, np.int64)]      # noqa: E731
    parts = [partition.by_assignment_deep(whole, owner, r, world, depth) for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate([p.owned_x for p in parts])), np.sort(whole.x_ids))
    for p in parts:      # a kept factor of more than two variables keeps every edge
        for f in p.model.factor_ids[p.model.factor_kind == 5]:
            assert (p.model.edge_fac == f).sum() == (whole.edge_fac == f).sum()
    sws = [OracleStateSweeper(p, 1e3) for p in parts]
    sweeps = 3 * depth + 2
    for k in range(sweeps):
        if k % depth == 0:
            for sw in sws:
                sw.pack()
            for r, p in enumerate(parts):
                for peer in p.peers:
                    back = [pp for pp in parts[peer.rank].peers if pp.rank == r][0]
                    sws[peer.rank].recv[back.recv] = sws[r].send[peer.send]
            for sw in sws:
                sw.unpack()
        for sw in sws:
            sw.sweep()
    g = flood_oracle_from_model(whole, 1e3)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    checked = 0
    for p, sw in zip(parts, sws):
        if len(p.owned_x) == 0:
            continue
        m, v = sw.g.marginals()
        li = np.searchsorted(sw.g.var_ids, p.owned_x); wi = np.searchsorted(g.var_ids, p.owned_x)
        assert np.array_equal(m[li], gm[wi], equal_nan=True) and np.array_equal(v[li], gv[wi], equal_nan=True)
        own = np.isin(sw.g.edge_var, p.owned_x)
        e = g.edge_index(sw.g.edge_var[own], sw.g.edge_fac[own])
        for name in ("f2v_m", "f2v_v"):
            assert np.array_equal(getattr(sw.g, name)[own], getattr(g, name)[e], equal_nan=True), name
        checked += int(own.sum())
    assert checked > 0
    if n_factors >= 500:      # (the small models are so well connected that a halo of depth 2 holds everything: the large one is really cut)
        assert any(p.model.n_edges < whole.n_edges for p in parts)


@pytest.mark.parametrize("depth", [2, 0])
def test_gloo_partitioned_convergence_with_the_residual_all_reduce(tmp_path, depth):
    """partition.converge over gloo: three ranks — deep halo (depth 2) or one message halo per sweep (depth 0, the
    HaloExchange.sweep(k) path) — sweep until the globally reduced residual is below 1e-12; all ranks stop after the same
    number of sweeps, and the owned marginal means then equal the dense solve of the whole grid (loopy Gaussian BP means
    are exact at convergence)."""
    from oracle import exact

    rows, cols, world = 4, 7, 3
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(rows), str(cols),
                                       "-1", out, str(depth)], env=env, cwd=ROOT))   # depth 0 is parsed as "no deep halo"
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    whole = cx.synth.gaussian_grid(rows * world, cols, seed=99)
    mean = exact.grid_posterior_mean(rows * world, cols, whole.meta["y"], whole.meta["r"], whole.meta["qh"], whole.meta["qv"]).ravel()
    runs = set()
    for r in range(world):
        n_run, res = np.load(out + f".rank{r}.conv.npy")
        runs.add(int(n_run))
        assert res <= 1e-12 and 0 < n_run < 4000
        d = np.load(out + f".rank{r}.npz")
        li = np.searchsorted(d["var_ids"], d["owned"])
        np.testing.assert_allclose(d["marg_m"][li], mean[d["owned"] - 1], rtol=1e-9)
    assert len(runs) == 1          # the all-reduce makes the decision common


@pytest.mark.parametrize("world,depth", [(3, 0), (3, 2), (2, 3), (8, 2)])
def test_gloo_strong_scaling_cut_of_one_grid(tmp_path, world, depth):
    """BASELINE config 4's shape: ONE grid cut into `world` near-equal row blocks (10 rows over 3 ranks: 3 / 3 / 4), as
    bench.py --gpus N does by default (partition.grid_rows / grid_rows_deep).  Owned messages and marginals equal the
    single-process sweep bit for bit."""
    rows, cols, sweeps = {3: 10, 2: 7, 8: 43}[world], 6, 8        # 8 ranks (what bench.py --gpus 8 runs): blocks of 5 / 6 rows, two-neighbour middle ranks
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(rows), str(cols),
                                       str(sweeps), out, str(depth), "strong"], env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    whole = cx.synth.gaussian_grid(rows, cols, seed=99)
    g = flood_oracle_from_model(whole, 1e6)
    g.sweep(sweeps)
    gm, gv = g.marginals()
    key = lambda v, f: v.astype(np.int64) * (1 << 32) + f  # noqa: E731
    gkey = key(g.edge_var, g.edge_fac)
    order = np.argsort(gkey)
    owned_all = []
    for r in range(world):
        d = np.load(out + f".rank{r}.npz")
        assert bool(d["audit_ok"]) and not bool(d["audit_bad"])
        own = np.isin(d["edge_var"], d["owned"])
        pos = order[np.searchsorted(gkey[order], key(d["edge_var"][own], d["edge_fac"][own]))]
        for name in ("f2v_m", "f2v_v"):
            a, b = d[name][own], getattr(g, name)[pos]
            assert np.array_equal(a, b, equal_nan=True), f"{name} differs on rank {r}"
        vi = np.searchsorted(g.var_ids, d["owned"])
        li = np.searchsorted(d["var_ids"], d["owned"])
        assert np.array_equal(d["marg_m"][li], gm[vi]) and np.array_equal(d["marg_v"][li], gv[vi])
        owned_all.append(d["owned"])
    assert np.array_equal(np.sort(np.concatenate(owned_all)), np.sort(whole.x_ids))
    b = partition._row_bounds(rows, world)
    assert [len(o) // cols for o in owned_all] == np.diff(b).tolist()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it: the parent spawns N ranks (before importing torch or touching
    a GPU), relays rank 0's line and returns 0; a rank that dies makes the whole run fail with its code.  --launch-check
    stops after the rendezvous, so this runs without a GPU."""
    import json

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"launcher": "ok", "world": 3, "sum_of_ranks_plus_one": 6}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"],
                       env=dict(env, CX_BENCH_FAIL_RANK="2"), cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 7
    assert "rank 2 exited with 7" in r.stderr


def _spawn(world, args):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")] + [str(a) for a in args], env=env, cwd=ROOT))
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize("world,T", [(2, 40), (3, 41), (4, 9)])
def test_gloo_chain_scan_partition_is_the_exact_smoother(tmp_path, world, T):
    """SURVEY §8e for chains: contiguous time blocks, ONE all-gather of the blocks' composed maps, a local pass
    (partition.ChainScanExchange over gloo, a numpy block standing in for the chain-scan handle): every rank's marginals are
    the exact posterior of the WHOLE chain (Thomas solve)."""
    from oracle import exact

    out = str(tmp_path / "res")
    _spawn(world, ["chain", T, out])
    whole = cx.synth.ssm_chain(T, seed=8, random_variances=True)
    em, ev = exact.ssm_chain_posterior(whole.data_y, whole.meta["r"], whole.meta["q"])
    seen = []
    for r in range(world):
        d = np.load(out + f".rank{r}.npz")
        idx = d["x"] - 1
        np.testing.assert_allclose(d["mean"], em[idx], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(d["var"], ev[idx], rtol=1e-10)
        seen.append(d["x"])
    assert np.array_equal(np.sort(np.concatenate(seen)), whole.x_ids)


@pytest.mark.parametrize("world,d,depth", [(2, 4, 2), (3, 2, 3)])
def test_gloo_deep_halo_for_d_dimensional_messages(tmp_path, world, d, depth):
    """dim > 1 partitions: time blocks of a d-dimensional chain with a deep halo (partition.contiguous_blocks(..., depth) carries
    edge roles and parameter sets; DeepHaloExchange moves (mean, covariance) rows), the C checker as the sweeper: owned marginals
    equal the single-process flooding sweeps bit for bit."""
    from oracle.mv import MvFloodC

    T, sweeps = 24, 3 * depth + 1
    out = str(tmp_path / "res")
    _spawn(world, ["mv", d, T, sweeps, depth, out])
    whole = cx.synth.lgssm_chain(T, d=d, seed=6)
    o = MvFloodC(whole)
    o.seed(0.0, 1e6)
    o.sweep(sweeps)
    m, S, ok = o.marginals()
    seen = []
    for r in range(world):
        g = np.load(out + f".rank{r}.npz")
        li = np.searchsorted(o.g.var_ids, g["owned"])
        assert ok[li].all() and g["ok"].all()
        assert np.array_equal(g["mean"], m[li]) and np.array_equal(g["cov"], S[li])
        seen.append(g["owned"])
    assert np.array_equal(np.sort(np.concatenate(seen)), np.sort(whole.x_ids))


@pytest.mark.parametrize("world,d,T", [(2, 2, 9), (3, 4, 31), (2, 64, 7)])
def test_gloo_chain_scan_partition_for_d_dimensional_chains(tmp_path, world, d, T):
    """SURVEY §8e for d-dimensional chains (round 3): contiguous time blocks, ONE all-gather of the blocks' composed linear-Gaussian
    maps, a local pass (partition.ChainScanExchange over gloo, a numpy block standing in for the dim > 1 chain-scan handle): every
    rank's marginals are the exact posterior of the WHOLE chain (block-tridiagonal solve).  d = 64 (round 4): the boundary messages are
    handed over AFTER the cut factor's rule, as the dim 64 handle takes them."""
    from oracle import exact

    out = str(tmp_path / "res")
    _spawn(world, ["mvchain", d, T, out])
    whole = cx.synth.lgssm_chain(T, d=d, seed=12)
    em, ecov = exact.lgssm_posterior(whole.data_y, whole.meta["A"], whole.meta["Q"], whole.meta["R"])
    seen = []
    for r in range(world):
        g = np.load(out + f".rank{r}.npz")
        idx = g["x"] - 1
        np.testing.assert_allclose(g["mean"], em[idx], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(g["cov"], ecov[idx], rtol=1e-9, atol=1e-12)
        seen.append(g["x"])
    assert np.array_equal(np.sort(np.concatenate(seen)), whole.x_ids)


def test_chain_scan_exchange_refuses_a_block_of_one_state():
    """world large relative to T leaves a rank with a single latent variable — no link, no block map: a clear error"""
    from cortex.jl_amd import partition

    model = cx.synth.ssm_chain(5, seed=1)
    part = partition.contiguous_blocks(model, 1, 4)
    with pytest.raises(ValueError, match="at least two"):
        partition.ChainScanExchange(None, part, None, None)
    partition.ChainScanExchange(None, partition.contiguous_blocks(model, 0, 2), None, None)     # 2 or 3 states per block: fine
