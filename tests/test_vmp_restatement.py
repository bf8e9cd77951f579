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


@pytest.mark.parametrize("n,seed", [(7, 3), (9, 5), (40, 4)])
def test_structured_requests_of_states_and_precisions_together(n, seed):
    """the reference's own experiment INCLUDING its last call, which names both precisions and every state in one request
    (test/inference_engine_tests.jl:1113): on the restated engine it equals three calls class by class — q(ssnoise), the states,
    q(obsnoise) — which is what StructuredVMP.update (and the device: cx_vmp.hip) does with it"""
    data = S.dataset(n, seed=seed)
    be = S.OracleBackend(S.structured_rule)
    arr = vmp.StructuredVMP(data)
    x = list(range(3, 3 + n))
    calls, mixed = [0], [0]

    def on_call(it, ids):
        which = vmp.which_of(ids, x, 1, 2)
        arr.update(which)
        mixed[0] += "x" in which and len(which) > 1
        xm, xw, ss, obs = _engine_state(be, x, 1, 2)
        np.testing.assert_allclose(np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs]), np.concatenate([xm, xw, ss, obs]), rtol=1e-11, atol=0,
                                   err_msg=f"n={n}: call {calls[0]} {which}")
        calls[0] += 1

    S.run_experiment(be, "structured", data, 5, on_call=on_call, calls_of=S.structured_calls)
    assert calls[0] == 5 * 13 and mixed[0] == 5


@pytest.mark.parametrize("seed", range(6))
def test_random_histories_of_structured_requests(seed):
    """random sequences of requests — single classes, both precisions, and states together with precisions in every order: wherever the
    array form accepts the request it equals the restated engine; where it refuses (x before ssnoise, q(obsnoise) not updated since
    the states were, no state update yet) the engine's result is none of the class-by-class orders, which is why it refuses"""
    import itertools
    n = 9
    rng = np.random.default_rng(seed)
    data = S.dataset(n, seed=seed + 10)
    be = S.OracleBackend(S.structured_rule)
    xs, y, obsn, ssn = S.make_ssm_model(be, n, S.structured_factor, None)
    for i in range(n):
        be.set_marginal(y[i], S.real(data[i]))
    arr = vmp.StructuredVMP(data)
    x = list(xs)
    menu = [x, [ssn], [obsn], [ssn, obsn], [ssn, obsn] + x, [obsn, ssn] + x, [ssn] + x + [obsn], [ssn] + x, [obsn] + x, x + [obsn], x + [ssn], x + [ssn, obsn]]
    accepted = refused = together = 0
    script = [x, [ssn], [obsn], [obsn, ssn] + x, [obsn], [ssn] + x + [obsn], [obsn] + x]      # a prefix every seed runs: three accepted joint requests
    for step in range(47):
        ids = script[step] if step < len(script) else menu[int(rng.integers(0, len(menu)))]
        which = vmp.which_of(ids, x, ssn, obsn)
        import copy
        trial = copy.deepcopy(arr)
        try:
            trial.update(which)
        except NotImplementedError:
            refused += 1
            continue              # (the engine is not asked either: its state would leave what the array form can follow)
        be.update_marginals(ids)
        arr = trial
        accepted += 1
        together += "x" in which and len(which) > 1
        xm, xw, ss, obs = _engine_state(be, x, ssn, obsn)
        np.testing.assert_allclose(np.concatenate([arr.xm, arr.xw, arr.ss, arr.obs]), np.concatenate([xm, xw, ss, obs]), rtol=1e-10, atol=0,
                                   err_msg=f"seed {seed}: request {which}")
    assert accepted >= 15 and refused >= 1 and together >= 3
