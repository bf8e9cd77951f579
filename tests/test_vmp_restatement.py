"""oracle/vmp.py (the array form of one update_marginals! call on the variational SSM models) pinned, call by call,
against the C restatement of the reference's engine driven through the transcribed resolvers and rules."""
import numpy as np
import pytest

from oracle import vmp
from tests import vmp_support as S


def _engine_state(be, x, ss, obs):
    xs = np.array([be.get_marginal(v)[1][:2] for v in x])
    return xs[:, 0], xs[:, 1], tuple(be.get_marginal(ss)[1][:2]), tuple(be.get_marginal(obs)[1][:2])


@pytest.mark.parametrize("kind,rule,cls,tol", [("mean_field", S.mean_field_rule, vmp.MeanFieldVMP, 0.0),
                                               ("structured", S.structured_rule, vmp.StructuredVMP, 1e-11)])
@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (7, 3), (40, 4)])
def test_array_form_equals_the_engine_call_by_call(kind, rule, cls, tol, n, seed):
    data = S.dataset(n, seed=seed)
    be = S.OracleBackend(rule)
    arr = cls(data)
    x = list(range(3, 3 + n))
    calls = [0]

    def on_call(it, ids):
        arr.update(vmp.which_of(ids, x, 1, 2))
        xm, xw, ss, obs = _engine_state(be, x, 1, 2)
        got = np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs])
        want = np.concatenate([xm, xw, ss, obs])
        if tol == 0.0:
            assert np.array_equal(got, want), f"{kind} n={n}: call {calls[0]} ({ids[:3]}...) differs"
        else:
            np.testing.assert_allclose(got, want, rtol=tol, atol=0, err_msg=f"{kind} n={n}: call {calls[0]}")
        calls[0] += 1

    S.run_experiment(be, kind, data, 5, on_call=on_call,
                     calls_of=S.mean_field_calls if kind == "mean_field" else S.structured_calls_by_class)
    assert calls[0] == 5 * (10 if kind == "mean_field" else 12)
