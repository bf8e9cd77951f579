"""Shared test plumbing: the same Model arrays go to the HIP path and to the CPU checker (oracle/)."""
import numpy as np

from oracle import ref


def flood_oracle_from_model(model, seed_variance=None):
    """FloodGraph (oracle/bp_flood.c) loaded with the model's data, mirroring synth.load_into_device; a synth.kary_model (factors of
    more than two variables) gets the k-ary checker (oracle/bp_kary.c)."""
    if model.meta.get("kind") == "kary":
        g = ref.KaryFloodGraph(model)
        if len(model.prior_var):
            g.set_message_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
        if len(model.data_var):
            g.set_data(model.data_var, model.data_fac, model.data_y)
        if seed_variance is not None:
            und = np.isnan(g.f2v_v) & g.kary_edge
            g.f2v_m[und], g.f2v_v[und] = 0.0, seed_variance
        return g
    g = ref.FloodGraph(model.edge_var, model.edge_fac, model.factor_ids, model.factor_var)
    if len(model.data_var):
        g.set_data(model.data_var, model.data_fac, model.data_y)
    if len(model.prior_var):
        g.set_message_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
    if seed_variance is not None:
        und = np.isnan(g.f2v_v) & (g.partner >= 0)
        g.f2v_m[und] = 0.0
        g.f2v_v[und] = seed_variance
    return g


def engine_oracle_from_model(model, trace=False):
    """The reference's InferenceEngine restated (oracle/cortex_ref.c) on the same graph and data."""
    n_nodes = int(max(model.edge_var.max(), model.factor_ids.max()))
    kind = np.zeros(n_nodes, dtype=np.int32)
    fkind = np.zeros(n_nodes, dtype=np.int32)
    p0 = np.ones(n_nodes)
    kind[np.unique(model.edge_var) - 1] = 1
    kind[model.factor_ids - 1] = 2
    fkind[model.factor_ids - 1] = np.where(model.factor_kind == 1, ref.F_GAUSS_ADD, ref.F_OPAQUE)
    p0[model.factor_ids - 1] = model.factor_var
    assert np.all(kind > 0), "ids must be dense for the restated BipartiteFactorGraph"
    E = ref.Engine(ref.P_SSM_BP, trace)
    E.bulk_build(kind, fkind, p0, model.edge_var, model.edge_fac)
    E.finalize()
    if len(model.data_var):
        E.set_messages_to_factor(model.data_var, model.data_fac, model.data_y, tag=ref.REAL)
    if len(model.prior_var):
        E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance, tag=ref.NORMAL)
    return E


def assert_close(actual, expected, rtol, what="", scale_by="median"):
    """|a-b| <= rtol * max(|b|, scale) elementwise, scale = typical magnitude of `expected`
    (means cross zero, so a pure relative test is ill-posed there); NaN patterns must agree."""
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, f"{what}: shape {actual.shape} vs {expected.shape}"
    na, ne = np.isnan(actual), np.isnan(expected)
    assert np.array_equal(na, ne), f"{what}: undefined-value pattern differs ({na.sum()} vs {ne.sum()} NaN)"
    ok = ~ne
    if not ok.any():
        return 0.0
    scale = max(float((np.median if scale_by == "median" else np.max)(np.abs(expected[ok]))), 1e-300)
    err = np.abs(actual[ok] - expected[ok]) / np.maximum(np.abs(expected[ok]), scale)
    worst = float(err.max())
    assert worst <= rtol, f"{what}: max rel err {worst:.3e} > {rtol:.1e}"
    return worst


def random_loopy_model(seed, world, nv=60, extra=25):
    """A connected random model with unary priors and `nv - 1 + extra` pairwise factors, plus a random variable→rank map."""
    import cortex.jl_amd as cx

    rng = np.random.default_rng(seed)
    pairs = set()
    for v in range(1, nv):                       # a spanning tree keeps the graph connected, extra edges make it loopy
        pairs.add((int(rng.integers(0, v)), v))
    while len(pairs) < nv - 1 + extra + 1:
        a, b = sorted(int(t) for t in rng.integers(0, nv, 2))
        if a != b:
            pairs.add((a, b))
    pairs = sorted(pairs)
    x = np.arange(1, nv + 1, dtype=np.int64)
    unary = nv + x
    pf = 2 * nv + 1 + np.arange(len(pairs), dtype=np.int64)
    pa = np.array([p[0] for p in pairs]) + 1
    pb = np.array([p[1] for p in pairs]) + 1
    whole = cx.synth.Model(edge_var=np.concatenate([x, pa, pb]), edge_fac=np.concatenate([unary, pf, pf]),
                           factor_ids=np.concatenate([unary, pf]),
                           factor_kind=np.concatenate([np.zeros(nv, np.int32), np.ones(len(pairs), np.int32)]),
                           factor_var=np.concatenate([np.ones(nv), rng.uniform(0.5, 2.0, len(pairs))]), x_ids=x,
                           prior_var=x, prior_fac=unary, prior_mean=rng.standard_normal(nv) * 2,
                           prior_variance=rng.uniform(0.5, 2.0, nv))
    owner_map = rng.integers(0, world, nv)
    return whole, (lambda ids: owner_map[np.asarray(ids, np.int64) - 1])


def mirror_engine_from_model(model, processor, trace=False):
    """The host mirror's InferenceEngine (cortex.jl_amd: the reference's Signals, resolver and scheduler, which the host keeps) on the
    graph of a scalar Model, ids as in the model (one dense 1-based id space shared by variables and factors)."""
    import cortex.jl_amd as cx

    n_nodes = int(max(model.edge_var.max(), model.factor_ids.max()))
    is_var = np.zeros(n_nodes + 1, bool)
    is_var[np.unique(model.edge_var)] = True
    fvar = {int(f): (int(k), float(q)) for f, k, q in zip(model.factor_ids, model.factor_kind, np.asarray(model.factor_var).reshape(len(model.factor_ids), -1)[:, 0])}
    graph = cx.BipartiteFactorGraph()
    for i in range(1, n_nodes + 1):
        if is_var[i]:
            got = graph.add_variable(cx.Variable(name="x", index=(i,)))
        else:
            kind, q = fvar[i]
            got = graph.add_factor(cx.Factor(functional_form=cx.GaussianAdditive(q) if kind == 1 else "prior"))
        assert got == i, "ids must be dense for the mirrored BipartiteFactorGraph"
    for v, f in zip(model.edge_var, model.edge_fac):
        graph.add_edge(int(v), int(f), cx.Connection(label="out"))
    return cx.InferenceEngine(model_engine=graph, inference_request_processor=processor, trace=trace)
