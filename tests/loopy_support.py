"""Call-by-call comparison of a processor behind the host mirror's scheduler with the restated reference engine (oracle/cortex_ref.c)
on LOOPY graphs: the same seeds, the same priors re-set before every call after the first (a user must, to make them fresh:
src/signal.jl:668-730), the same requests — then the execution ORDER (variants, one by one) and every message and marginal."""
import numpy as np

import cortex.jl_amd as cx
from cortex.jl_amd import get_value, update_marginals
from oracle import ref
from tests.helpers import assert_close, engine_oracle_from_model, mirror_engine_from_model, random_loopy_model

SEED_VARIANCE = 1e6


def models():
    rnd, _ = random_loopy_model(11, 1, nv=60, extra=25)
    return {"grid8x9": cx.synth.gaussian_grid(8, 9, seed=5), "grid48x40": cx.synth.gaussian_grid(48, 40, seed=1), "random": rnd}


def pairwise_edges(model):
    pw = set(int(f) for f, k in zip(model.factor_ids, model.factor_kind) if k == 1)
    keep = np.array([int(f) in pw for f in model.edge_fac])
    return model.edge_var[keep], model.edge_fac[keep]


def variant_key(variant):
    V = cx.InferenceSignalVariants
    if isinstance(variant, V.MessageToVariable):
        return (ref.VAR_MSG_TO_VARIABLE, int(variant.variable_id), int(variant.factor_id), 0, 0)
    if isinstance(variant, V.MessageToFactor):
        return (ref.VAR_MSG_TO_FACTOR, int(variant.variable_id), int(variant.factor_id), 0, 0)
    if isinstance(variant, V.IndividualMarginal):
        return (ref.VAR_MARGINAL, int(variant.variable_id), 0, 0, 0)
    if isinstance(variant, V.ProductOfMessages):
        return (ref.VAR_PRODUCT, int(variant.variable_id), 0, int(variant.range[0]), int(variant.range[1]))
    raise TypeError(variant)


def oracle_key(E, s):
    k, v, f, lo, hi = E.variant(s)
    return (k, v, f if k in (ref.VAR_MSG_TO_VARIABLE, ref.VAR_MSG_TO_FACTOR) else 0, lo, hi)


class HostBackend:
    """a processor that computes on the host with the reference's own arithmetic (tests/test_host_mirror.py)"""

    def __init__(self):
        from tests.test_host_mirror import NMV, SSMBeliefPropagationProcessor

        self.NMV = NMV
        self.proc = SSMBeliefPropagationProcessor()
        self.log = []
        inner = self.proc.process

        def process(engine, variable_id, dependency):
            self.log.append(dependency.variant)
            return inner(engine, variable_id, dependency)

        self.proc.process = process

    def bind(self, engine, model):
        self.engine = engine

    def set_message_to_variable(self, v, f, mean, variance):
        cx.set_value(self.engine.get_connection_message_to_variable(int(v), int(f)), self.NMV(float(mean), float(variance)))

    def messages(self, model, to_variable):
        out = np.full((len(model.edge_var), 2), np.nan)
        for i, (v, f) in enumerate(zip(model.edge_var, model.edge_fac)):
            c = self.engine.get_connection(int(v), int(f))
            val = get_value(cx.get_connection_message_to_variable(c) if to_variable else cx.get_connection_message_to_factor(c))
            if not isinstance(val, cx.UndefValue):
                out[i] = (val.mean, val.variance)
        return out

    def marginals(self, ids):
        vals = [get_value(cx.get_variable_marginal(self.engine.get_variable(int(v)))) for v in ids]
        return np.array([[a.mean, a.variance] for a in vals])


class HipBackend:
    """HipProcessor: values on the device, rule calls as launches through the C ABI"""

    def __init__(self, mode, schedule=None):
        self.proc = cx.HipProcessor(mode=mode) if schedule is None else cx.HipProcessor(mode=mode, schedule=schedule)
        self.log = self.proc.execution_log

    def bind(self, engine, model):
        self.engine = engine

    def set_message_to_variable(self, v, f, mean, variance):
        self.proc.set_value(self.engine.get_connection_message_to_variable(int(v), int(f)), cx.NormalMeanVariance(float(mean), float(variance)))

    def messages(self, model, to_variable):
        from cortex.jl_amd import _lib as L

        return self.proc.dev.get_messages(model.edge_var, model.edge_fac, L.TO_VARIABLE if to_variable else L.TO_FACTOR)

    def marginals(self, ids):
        return self.proc.dev.get_marginals(ids)


def run_calls(model, backend, n_calls=3, request=None, rtol=1e-9):
    request = model.x_ids if request is None else np.asarray(request)
    E = engine_oracle_from_model(model, trace=True)
    engine = mirror_engine_from_model(model, backend.proc)
    backend.bind(engine, model)
    pv, pf = pairwise_edges(model)
    E.set_messages_to_variable(pv, pf, np.zeros(len(pv)), np.full(len(pv), SEED_VARIANCE))
    for v, f in zip(pv, pf):
        backend.set_message_to_variable(v, f, 0.0, SEED_VARIANCE)
    for v, f, m, s in zip(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance):
        backend.set_message_to_variable(v, f, m, s)
    n_executed = []
    for call in range(n_calls):
        if call:
            E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
            for v, f, m, s in zip(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance):
                backend.set_message_to_variable(v, f, m, s)
        backend.log.clear()
        update_marginals(engine, [int(v) for v in request])
        E.update_marginals(request)
        want_order = [oracle_key(E, s) for _r, _v, s, _b, _a in E.trace()]
        got_order = [variant_key(v) for v in backend.log]
        assert got_order == want_order, f"call {call + 1}: execution order differs from the restated reference engine"
        n_executed.append(len(got_order))
        for to_variable in (True, False):
            tags, a, b = E.get_messages(model.edge_var, model.edge_fac, to_variable)
            got = backend.messages(model, to_variable)
            und = tags == ref.UNDEF
            name = "f2v" if to_variable else "v2f"
            assert np.array_equal(np.isnan(got[:, 1]), und), f"call {call + 1}: the same {name} messages are defined"
            assert_close(got[~und, 0], a[~und], rtol, f"call {call + 1} {name} mean")
            assert_close(got[~und, 1], b[~und], rtol, f"call {call + 1} {name} variance")
        _tags, em, ev = E.get_marginals(request)
        marg = backend.marginals(request)
        assert_close(marg[:, 0], em, rtol, f"call {call + 1} marginal mean")
        assert_close(marg[:, 1], ev, rtol, f"call {call + 1} marginal variance")
    return n_executed
