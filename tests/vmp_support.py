"""Test support: the two variational SSM models of the reference's test-suite, written ONCE against a small back-end
protocol and run on both restatements of the reference's engine:

  * OracleBackend — oracle/cortex_ref.c (C restatement of Signal / scheduler) with the rules below supplied as a callback
  * MirrorBackend — the product's host-side mirror (cortex.jl_amd/{signal,inference_engine,...}.py)

Transcribed from test/inference_engine_tests.jl:593-805 ("Mean Field") and :807-1147 ("Structured"); the value types and
their products follow test/runtests.jl:48-93 (NormalMeanPrecision, Gamma, MvNormalMeanPrecision) operation by
operation.  Values are (tag, payload) pairs with the oracle's tags."""
import numpy as np

from oracle import ref

REAL, NORMAL_MP, GAMMA, MVN2 = ref.REAL, ref.NORMAL_MP, ref.GAMMA, ref.MVNORMAL2
K_M2F, K_M2V, K_PROD, K_MARG, K_JOINT = ref.VAR_MSG_TO_FACTOR, ref.VAR_MSG_TO_VARIABLE, ref.VAR_PRODUCT, ref.VAR_MARGINAL, ref.VAR_JOINT


# ---- distributions (test/runtests.jl:48-93) ------------------------------------------------------------------------
def normal_mp(mean, precision):
    return (NORMAL_MP, [float(mean), float(precision)])


def gamma(shape, scale):
    return (GAMMA, [float(shape), float(scale)])


def real(y):
    return (REAL, [float(y)])


def mean(v):
    tag, p = v
    if tag == GAMMA:
        return p[0] * p[1]                      # shape * scale, runtests.jl:66
    if tag == MVN2:
        return [p[0], p[1]]
    return p[0]


def var(v):
    tag, p = v
    if tag == NORMAL_MP:
        return 1 / p[1]                         # runtests.jl:54
    if tag == GAMMA:
        return p[0] * p[1] ** 2
    raise TypeError(tag)


def precision(v):
    tag, p = v
    assert tag == NORMAL_MP
    return p[1]


def product(left, right):
    if left[0] == NORMAL_MP and right[0] == NORMAL_MP:       # runtests.jl:78-84
        xi = left[1][0] * left[1][1] + right[1][0] * right[1][1]
        w = left[1][1] + right[1][1]
        return normal_mp((1 / w) * xi, w)
    if left[0] == GAMMA and right[0] == GAMMA:                # runtests.jl:86-88
        return gamma(left[1][0] + right[1][0] - 1, (left[1][1] * right[1][1]) / (left[1][1] + right[1][1]))
    raise TypeError(f"product of {left[0]} and {right[0]}")


def fold(values):
    acc = values[0]
    for v in values[1:]:
        acc = product(acc, v)
    return acc


def gamma_from_spread(spread):
    theta = 2 / spread
    return gamma(1.5, theta)


# ---- rules -----------------------------------------------------------------------------------------------------------
def mean_field_rule(api, sig):
    """SSMMeanFieldInferenceRequestProcessor, test/inference_engine_tests.jl:633-689"""
    kind, _var, _fac = api.variant(sig)
    deps = api.dependencies(sig)
    vals = [api.value(d) for d in deps]
    if kind in (K_MARG, K_M2F, K_PROD):
        return fold(vals)
    assert kind == K_M2V and len(deps) == 2
    names = [api.variable_name(api.variant(d)[1]) for d in deps]

    def first(name):
        return names.index(name) if name in names else None

    x, y, ss, obs = first("x"), first("y"), first("ssnoise"), first("obsnoise")
    if x is not None and ss is not None:
        return normal_mp(mean(vals[x]), mean(vals[ss]))
    if y is not None and obs is not None:
        return normal_mp(mean(vals[y]), mean(vals[obs]))
    if y is not None and x is not None:
        q_out, q_mu = mean(vals[y]), vals[x]
        return gamma_from_spread(var(q_mu) + (q_out - mean(q_mu)) ** 2)
    if names.count("x") == 2:
        q_out, q_mu = vals[0], vals[1]
        return gamma_from_spread(var(q_out) + var(q_mu) + (mean(q_out) - mean(q_mu)) ** 2)
    raise RuntimeError("Unreachable reached")


def structured_rule(api, sig):
    """SSMStructuredInferenceRequestProcessor, test/inference_engine_tests.jl:905-1030"""
    kind, _var, fac = api.variant(sig)
    deps = api.dependencies(sig)
    vals = [api.value(d) for d in deps]
    kinds = [api.variant(d)[0] for d in deps]
    if kind in (K_MARG, K_M2F, K_PROD):
        return fold(vals)
    if kind == K_JOINT:                                       # :939-967
        assert len(deps) == 3 and kinds == [K_M2F, K_M2F, K_MARG]
        m1, m2, mrg = vals
        xi_out, w_out = precision(m1) * mean(m1), precision(m1)
        xi_mu, w_mu = precision(m2) * mean(m2), precision(m2)
        w_bar = mean(mrg)
        W = np.array([[w_out + w_bar, -w_bar], [-w_bar, w_mu + w_bar]])
        mu = np.linalg.inv(W) @ np.array([xi_out, xi_mu])
        return (MVN2, [mu[0], mu[1], W[0, 0], W[0, 1], W[1, 0], W[1, 1]])
    assert kind == K_M2V
    form = api.factor_form(fac)
    if form == "likelihood":                                  # :978-997
        names = [api.variable_name(api.variant(d)[1]) for d in deps]

        def first(name):
            return names.index(name) if name in names else None

        x, y, obs = first("x"), first("y"), first("obsnoise")
        if y is not None and obs is not None:
            return normal_mp(mean(vals[y]), mean(vals[obs]))
        if x is not None and y is not None:
            q_out, q_mu = mean(vals[y]), vals[x]
            return gamma_from_spread(var(q_mu) + (q_out - mean(q_mu)) ** 2)
        raise RuntimeError("unreachable reached in likelihood")
    assert form == "transition"                               # :998-1024
    msg = kinds.index(K_M2F) if K_M2F in kinds else None
    mrg = kinds.index(K_MARG) if K_MARG in kinds else None
    jm = kinds.index(K_JOINT) if K_JOINT in kinds else None
    if msg is not None and mrg is not None:
        return normal_mp(mean(vals[msg]), 1 / (var(vals[msg]) + 1 / mean(vals[mrg])))
    if jm is not None:
        p = vals[jm][1]
        m = p[:2]
        V = np.linalg.inv(np.array([[p[2], p[3]], [p[4], p[5]]]))
        return gamma_from_spread(V[0, 0] - V[0, 1] - V[1, 0] + V[1, 1] + (m[0] - m[1]) ** 2)
    raise RuntimeError("unreachable reached")


# ---- dependency resolvers --------------------------------------------------------------------------------------------
def mean_field_variable(api, variable_id):
    """MeanFieldResolver, :599-607"""
    marginal = api.marginal(variable_id)
    for f in api.connected_factors(variable_id):
        api.add_dependency(marginal, api.message_to_variable(variable_id, f), intermediate=True)


def mean_field_factor(api, factor_id):
    """MeanFieldResolver, :609-621 (also the likelihood branch of the structured resolver, :823-834)"""
    ids = api.connected_variables(factor_id)
    for v1 in ids:
        for v2 in ids:
            if v1 != v2:
                api.add_dependency(api.message_to_variable(v1, factor_id), api.marginal(v2), weak=True)


def structured_factor(api, factor_id):
    """StructuredResolver, :816-897"""
    if api.factor_form(factor_id) == "likelihood":
        return mean_field_factor(api, factor_id)
    ids = api.connected_variables(factor_id)
    clusters = {}
    for v in ids:                                             # clusters by variable name, :840-848
        clusters.setdefault(api.variable_name(v), []).append(v)
    deps = []
    for cluster in clusters.values():
        if len(cluster) == 1:
            deps.append(api.marginal(cluster[0]))
        else:
            joint = api.new_joint_marginal(factor_id, cluster)
            for v in cluster:
                api.link_signal_to_variable(v, joint)
                api.add_local_marginal_to_factor(factor_id, joint)
                api.add_dependency(joint, api.message_to_factor(v, factor_id), weak=True)
            deps.append(joint)
    for d1 in deps:                                           # :872-876
        for d2 in deps:
            if api.is_joint(d1) and not api.same(d1, d2):
                api.add_dependency(d1, d2, weak=True)
    for index, cluster in enumerate(clusters.values()):      # :878-895
        for m1 in cluster:
            for m2 in cluster:
                if m1 != m2:
                    api.add_dependency(api.message_to_variable(m1, factor_id), api.message_to_factor(m2, factor_id))
        for m1 in cluster:
            for another_index, other in enumerate(deps):
                if index != another_index:
                    api.add_dependency(api.message_to_variable(m1, factor_id), other, weak=True)


# ---- back-ends -------------------------------------------------------------------------------------------------------
class OracleBackend:
    """oracle/cortex_ref.c: C scheduler and readiness bits, rules through the callback processor."""

    def __init__(self, rule):
        self.E = ref.Engine(ref.P_CALLBACK)
        self.names, self.forms, self.joints = {}, {}, set()
        self.E.set_rule(lambda s: rule(self, s))

    # graph
    def add_variable(self, name):
        v = self.E.add_variable()
        self.names[v] = name
        return v

    def add_factor(self, form):
        f = self.E.add_factor()
        self.forms[f] = form
        return f

    def add_edge(self, v, f):
        self.E.add_edge(v, f)

    def resolve(self, factor_resolver, variable_resolver):
        """resolve_dependencies!, dependencies.jl:5-15: factors first, then variables"""
        self.E.finalize(resolve_dependencies=False)
        for f in self.E.factor_ids():
            factor_resolver(self, int(f))
        for v in self.E.variable_ids():
            if variable_resolver is None:
                self.E.resolve_variable_default(int(v))
            else:
                variable_resolver(self, int(v))

    # resolver protocol
    def marginal(self, v): return self.E.marginal(v)
    def message_to_variable(self, v, f): return self.E.message_to_variable(v, f)
    def message_to_factor(self, v, f): return self.E.message_to_factor(v, f)
    def connected_variables(self, f): return self.E.neighbors(f)
    def connected_factors(self, v): return self.E.neighbors(v)
    def variable_name(self, v): return self.names[v]
    def factor_form(self, f): return self.forms[f]
    def add_dependency(self, s, d, weak=False, intermediate=False): self.E.add_dependency(s, d, weak=weak, intermediate=intermediate)
    def link_signal_to_variable(self, v, s): self.E.link_signal_to_variable(v, s)
    def add_local_marginal_to_factor(self, f, s): pass        # bookkeeping only in the reference (model_engine.jl:150-153)
    def is_joint(self, s): return s in self.joints
    def same(self, a, b): return a == b

    def new_joint_marginal(self, factor_id, cluster):
        s = self.E.signal()
        self.E.set_variant(s, K_JOINT, cluster[0], factor_id, cluster[0], cluster[-1])
        self.joints.add(s)
        return s

    # rule protocol
    def variant(self, s):
        k, v, f, _lo, _hi = self.E.variant(s)
        return k, v, f

    def dependencies(self, s): return self.E.dependencies(s)

    def value(self, s):
        tag, six = self.E.get_value_ex(s)
        return (tag, six)

    # user protocol
    def set_marginal(self, v, value): self.E.set_value_ex(self.E.marginal(v), value[0], value[1])
    def get_marginal(self, v): return self.value(self.E.marginal(v))
    def update_marginals(self, ids): self.E.update_marginals(ids)


class MirrorBackend:
    """The product's host-side mirror of the reference API."""

    def __init__(self, rule):
        import cortex.jl_amd as cx
        from cortex.jl_amd import model_engine
        self.cx, self.me = cx, model_engine
        self.graph = cx.BipartiteFactorGraph()
        self.rule = rule
        self.engine = None

    def add_variable(self, name): return self.graph.add_variable(self.cx.Variable(name=name))
    def add_factor(self, form): return self.graph.add_factor(self.cx.Factor(functional_form=form))
    def add_edge(self, v, f): self.graph.add_edge(v, f, self.cx.Connection(label="out"))

    def resolve(self, factor_resolver, variable_resolver):
        cx, be = self.cx, self

        class Resolver(cx.AbstractDependencyResolver):
            def resolve_factor_dependencies(self, engine, factor_id):
                be.engine = engine
                factor_resolver(be, factor_id)

            def resolve_variable_dependencies(self, engine, variable_id):
                be.engine = engine
                if variable_resolver is None:
                    cx.DefaultDependencyResolver().resolve_variable_dependencies(engine, variable_id)
                else:
                    variable_resolver(be, variable_id)

        class Processor(cx.AbstractInferenceRequestProcessor):
            def _any(self, engine, variant, signal, dependencies):
                return be.rule(be, signal)
            compute_message_to_variable = compute_message_to_factor = compute_individual_marginal = _any
            compute_product_of_messages = compute_joint_marginal = _any

        self.engine = cx.InferenceEngine(model_engine=self.graph, dependency_resolver=Resolver(),
                                         inference_request_processor=Processor())

    def marginal(self, v): return self.cx.get_variable_marginal(self.engine.get_variable(v))
    def message_to_variable(self, v, f): return self.engine.get_connection_message_to_variable(v, f)
    def message_to_factor(self, v, f): return self.engine.get_connection_message_to_factor(v, f)
    def connected_variables(self, f): return self.engine.get_connected_variable_ids(f)
    def connected_factors(self, v): return self.engine.get_connected_factor_ids(v)
    def variable_name(self, v): return self.engine.get_variable(v).name
    def factor_form(self, f): return self.cx.get_factor_functional_form(self.engine.get_factor(f))
    def add_dependency(self, s, d, weak=False, intermediate=False): self.cx.add_dependency(s, d, weak=weak, intermediate=intermediate)
    def link_signal_to_variable(self, v, s): self.cx.link_signal_to_variable(self.engine.get_variable(v), s)
    def add_local_marginal_to_factor(self, f, s): self.me.add_local_marginal_to_factor(self.engine.get_factor(f), s)
    def is_joint(self, s): return self.cx.isa_variant(s, self.cx.InferenceSignalVariants.JointMarginal)
    def same(self, a, b): return a is b

    def new_joint_marginal(self, factor_id, cluster):
        s = self.cx.create_inference_signal()
        self.cx.set_variant(s, self.cx.InferenceSignalVariants.JointMarginal(factor_id, tuple(cluster)))
        return s

    def variant(self, s):
        V, v = self.cx.InferenceSignalVariants, self.cx.get_variant(s)
        if isinstance(v, V.MessageToFactor): return K_M2F, v.variable_id, v.factor_id
        if isinstance(v, V.MessageToVariable): return K_M2V, v.variable_id, v.factor_id
        if isinstance(v, V.IndividualMarginal): return K_MARG, v.variable_id, 0
        if isinstance(v, V.ProductOfMessages): return K_PROD, v.variable_id, 0
        if isinstance(v, V.JointMarginal): return K_JOINT, v.variable_ids[0], v.factor_id
        raise TypeError(v)

    def dependencies(self, s): return list(self.cx.get_dependencies(s))
    def value(self, s): return self.cx.get_value(s)
    def set_marginal(self, v, value): self.cx.set_value(self.marginal(v), value)
    def get_marginal(self, v): return self.value(self.marginal(v))
    def update_marginals(self, ids): self.cx.update_marginals(self.engine, list(ids) if isinstance(ids, (list, tuple)) else ids)


# ---- the model and the experiments -------------------------------------------------------------------------------------
def make_ssm_model(be, n, factor_resolver, variable_resolver):
    """make_ssm_model, :691-729 / :1032-1071 (identical graphs; ids: ssnoise 1, obsnoise 2, x 3.., y n+3.., factors after)"""
    ssnoise, obsnoise = be.add_variable("ssnoise"), be.add_variable("obsnoise")
    x = [be.add_variable("x") for _ in range(n)]
    y = [be.add_variable("y") for _ in range(n)]
    likelihood = [be.add_factor("likelihood") for _ in range(n)]
    transition = [be.add_factor("transition") for _ in range(n - 1)]
    for i in range(n):
        be.add_edge(y[i], likelihood[i]); be.add_edge(x[i], likelihood[i]); be.add_edge(obsnoise, likelihood[i])
    for i in range(n - 1):
        be.add_edge(x[i], transition[i]); be.add_edge(x[i + 1], transition[i]); be.add_edge(ssnoise, transition[i])
    be.resolve(factor_resolver, variable_resolver)
    be.set_marginal(ssnoise, gamma(1.0, 1.0))
    be.set_marginal(obsnoise, gamma(1.0, 1.0))
    for i in range(n):
        be.set_marginal(x[i], normal_mp(0.0, 1.0))
    return x, y, obsnoise, ssnoise


def mean_field_calls(x, ssnoise, obsnoise, iteration):
    """the update_marginals! calls of one VMP iteration, :740-765"""
    calls = [x, [ssnoise], [obsnoise]] if iteration // 2 == 0 else [[obsnoise], [ssnoise], x]
    calls += [[obsnoise]] * 3 + [[ssnoise]] * 3 + [[ssnoise, obsnoise]]
    return calls


def structured_calls(x, ssnoise, obsnoise, iteration):
    """:1081-1107"""
    calls = [x, [ssnoise], [obsnoise]] if iteration // 2 == 1 else [[obsnoise], [ssnoise], x]
    calls += [[ssnoise]] * 3 + [x] * 2 + [[obsnoise]] * 3 + [[ssnoise, obsnoise], [ssnoise, obsnoise] + list(x)]
    return calls


def dataset(n, seed=1234, ssnoise_real=100.0, obsnoise_real=100.0):
    """random walk + observations, :775-785 (StableRNG streams cannot be regenerated; the asserted facts are inequalities)"""
    rng = np.random.default_rng(seed)
    walk = [0.0]
    for _ in range(1, n):
        walk.append(walk[-1] + rng.standard_normal() / np.sqrt(ssnoise_real))
    return [w + rng.standard_normal() / np.sqrt(obsnoise_real) for w in walk]


def structured_calls_by_class(x, ssnoise, obsnoise, iteration):
    """structured_calls without its last request, which names x TOGETHER with the precisions: there the order of
    evaluation is emergent from the lazy readiness flags (q(ssnoise) is computed on the fly when the first chain message
    finds it pending, q(obsnoise) before or after the x marginals depending on what changed since its last update).
    Engine-level tests cover it (both restatements, bit for bit); the array form and the device update class by class."""
    return structured_calls(x, ssnoise, obsnoise, iteration)[:-1]


def run_experiment(be, kind, data, vmp_iterations, on_call=None, calls_of=None):
    n = len(data)
    if kind == "mean_field":
        x, y, obsnoise, ssnoise = make_ssm_model(be, n, mean_field_factor, mean_field_variable)
        calls_of = calls_of or mean_field_calls
    else:
        x, y, obsnoise, ssnoise = make_ssm_model(be, n, structured_factor, None)
        calls_of = calls_of or structured_calls
    for i in range(n):
        be.set_marginal(y[i], real(data[i]))
    for it in range(1, vmp_iterations + 1):
        for ids in calls_of(x, ssnoise, obsnoise, it):
            be.update_marginals(ids)
            if on_call is not None:
                on_call(it, ids)
    return {"x": [be.get_marginal(v) for v in x], "ssnoise": be.get_marginal(ssnoise), "obsnoise": be.get_marginal(obsnoise),
            "ids": (x, y, obsnoise, ssnoise)}
