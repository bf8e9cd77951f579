"""partition.by_components (independent objects): the trees of a forest / the chains of a batch dealt to the ranks, no
exchange afterwards.  CPU: the sub-models cover the model exactly once, balance the edges, and — components being independent — each
sub-model's dense posterior IS the whole model's posterior of its variables.  The device side: tests/test_gpu_partition.py."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from cortex.jl_amd import partition
from tests.kary_support import dense_posterior


@pytest.mark.parametrize("shape,components,world", [("random", 7, 3), ("deep", 12, 8), ("comb", 5, 2), ("star", 3, 4)])
def test_forest_components_cover_the_model_once_and_keep_its_posterior(shape, components, world):
    m = cx.synth.tree_model(240, seed=17, shape=shape, components=components, observe=0.3)
    ids, em, ev = dense_posterior(m)
    full = {int(i): (a, b) for i, a, b in zip(ids, em, ev)}
    seen, loads, facs = set(), [], []
    for r in range(world):
        sub = partition.by_components(m, r, world)
        loads.append(len(sub.edge_var))
        facs += [int(f) for f in sub.factor_ids]
        if not len(sub.edge_var):
            continue                                     # more ranks than components: an empty share
        i2, e2, v2 = dense_posterior(sub)
        mine = set(int(i) for i in i2)
        assert not (mine & seen), "a variable on two ranks"
        seen |= mine
        assert set(int(v) for v in sub.x_ids) == mine
        np.testing.assert_allclose(e2, [full[int(i)][0] for i in i2], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(v2, [full[int(i)][1] for i in i2], rtol=1e-12)
    assert seen == set(full) and sum(loads) == len(m.edge_var)
    assert sorted(facs) == sorted(int(f) for f in m.factor_ids), "every factor on exactly one rank"
    if components >= 2 * world:
        assert max(loads) <= 2.5 * (sum(loads) / world), "largest-first dealing keeps the shares comparable"


def test_a_batch_of_d_dimensional_chains():
    chains = [cx.synth.lgssm_chain(T, d=4, seed=30 + T) for T in (40, 7, 90, 33, 12)]
    m = cx.synth.concat_models(chains)
    world = 2
    subs = [partition.by_components(m, r, world) for r in range(world)]
    assert sum(len(s.edge_var) for s in subs) == len(m.edge_var)
    assert np.array_equal(np.sort(np.concatenate([s.x_ids for s in subs])), np.sort(m.x_ids))
    # whole chains stay together: every sub-model's latent variables are a union of whole chains
    off = 0
    for c in chains:
        ids = set(int(v) for v in m.x_ids[off:off + len(c.x_ids)])
        off += len(c.x_ids)
        assert sum(ids <= set(int(v) for v in s.x_ids) for s in subs) == 1
    assert abs(len(subs[0].edge_var) - len(subs[1].edge_var)) <= 4 * 33
    assert all(len(s.data_var) == len(s.x_ids) and s.dim == 4 and s.psets is m.psets for s in subs)
