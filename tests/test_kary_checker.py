"""CPU: oracle/bp_kary.c (the k-ary linear-Gaussian factor rule in moment form) pinned by mathematics: on a tree the fixed point of
sum-product equals the marginals of the joint Gaussian — here a dense solve that shares nothing with message passing.  The reference
has no such factor (it wires the dependencies, src/dependencies.jl:17-31, and leaves the rule to the user): parity unpinned by
anything of the reference's, stated in the checker's header."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from oracle import ref
from tests.kary_support import dense_posterior


@pytest.mark.parametrize("n_factors,seed,observe", [(1, 1, 0.0), (3, 2, 0.0), (12, 3, 0.0), (40, 4, 0.3), (25, 5, 0.6)])
def test_tree_fixed_point_equals_the_dense_posterior(n_factors, seed, observe):
    m = cx.synth.kary_model(n_factors, seed=seed, tree=True, observe=observe)
    g = ref.KaryFloodGraph(m)
    g.set_message_to_variable(m.prior_var, m.prior_fac, m.prior_mean, m.prior_variance)
    if len(m.data_var):
        g.set_data(m.data_var, m.data_fac, m.data_y)
    g.sweep(2 * n_factors + 4)
    mm, vv = g.marginals()
    ids, em, ev = dense_posterior(m)
    idx = np.searchsorted(g.var_ids, ids)
    np.testing.assert_allclose(mm[idx], em, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(vv[idx], ev, rtol=1e-9, atol=1e-12)


def test_every_edge_message_depends_on_all_other_edges_of_its_factor():
    """the wiring of dependencies.jl:17-31: a message out of a factor is defined exactly when every OTHER message into it is"""
    m = cx.synth.kary_model(1, seed=9, k_choices=(4,))
    g = ref.KaryFloodGraph(m)
    g.set_message_to_variable(m.prior_var[:-1], m.prior_fac[:-1], m.prior_mean[:-1], m.prior_variance[:-1])     # one variable hears nothing
    g.sweep(3)
    silent = int(m.prior_var[-1])
    k_edges = np.flatnonzero(g.kary_edge)
    for e in k_edges:
        v = int(g.edge_var[e])
        # the silent variable's own message to the factor is undefined, so every OTHER edge's message out of the factor is undefined too;
        # the message TO the silent variable needs only the others and is defined
        assert np.isnan(g.f2v_v[e]) == (v != silent)
