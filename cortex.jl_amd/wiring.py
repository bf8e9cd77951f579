"""Dependency wirings for cx_graph_wire as arrays: the variational resolvers of the reference's test-suite, vectorised over any graph of
CX_FACTOR_NORMAL_PRECISION factors (out ~ N(in, 1 / precision)) and opaque prior factors.

A user of the reference writes an `AbstractDependencyResolver` whose two methods call add_dependency! signal by signal
(/root/reference/src/dependencies.jl:1-15); cx_graph_wire takes those calls as triples.  The two resolvers the reference's tests define
(/root/reference/test/inference_engine_tests.jl:597-621 "MeanFieldResolver", :810-897 "StructuredResolver") make the same calls for every
factor of a kind, so their triples can be built with array operations — a model of 10^6 factors is wired in a second, where a Python loop over
add_dependency calls takes minutes.  Per signal the dependencies come out in the order the resolvers add them (that order is what the
scheduler walks, /root/reference/src/signal.jl:466-490); the order of the calls ACROSS signals only orders listener lists, which nothing
reads in order.

Triples are rows (kind, variable_id, factor_id) as DeviceGraph.graph_wire takes them."""
from typing import NamedTuple

import numpy as np

from . import _lib as L


class Triples(NamedTuple):
    signals: np.ndarray        # (N, 3) int64: kind, variable id, factor id
    dependencies: np.ndarray   # (N, 3) int64
    flags: np.ndarray          # (N,) int32


def _rows(kind, var, fac):
    var = np.asarray(var, dtype=np.int64)
    out = np.empty((len(var), 3), dtype=np.int64)
    out[:, 0] = kind; out[:, 1] = var; out[:, 2] = fac
    return out


def _by_factor(edge_var, edge_fac, edge_role):
    """the three variables of every NORMAL_PRECISION factor by role: factor ids (ascending), out, in, precision"""
    edge_var, edge_fac, edge_role = np.asarray(edge_var, np.int64), np.asarray(edge_fac, np.int64), np.asarray(edge_role, np.int64)
    order = np.lexsort((edge_role, edge_fac))
    f, v, r = edge_fac[order], edge_var[order], edge_role[order]
    if len(f) % 3 or not (np.array_equal(r[0::3], np.full(len(f) // 3, L.ROLE_OUT)) and np.array_equal(r[1::3], np.full(len(f) // 3, L.ROLE_IN))
                          and np.array_equal(r[2::3], np.full(len(f) // 3, L.ROLE_PRECISION)) and np.array_equal(f[0::3], f[2::3])):
        raise ValueError("every factor needs exactly three edges with roles OUT, IN, PRECISION")
    return f[0::3], v[0::3], v[1::3], v[2::3]


def _mean_field_factors(fac, out, inn, prec):
    """MeanFieldResolver.resolve_factor_dependencies! (:609-621): every message out of a factor depends WEAKLY on the marginals of the factor's
    two other variables, in ascending variable id order"""
    sig, dep = [], []
    trio = np.stack([out, inn, prec], axis=1)
    trio.sort(axis=1)                                   # the reference iterates get_connected_variable_ids: ascending
    for target in range(3):
        for other in range(3):
            if other != target:
                sig.append(_rows(L.ITEM_MESSAGE_TO_VARIABLE, trio[:, target], fac))
                dep.append(_rows(L.ITEM_INDIVIDUAL_MARGINAL, trio[:, other], 0))
    # per signal the two dependencies are appended in ascending id order because `other` runs ascending for a fixed target
    return np.concatenate(sig), np.concatenate(dep), np.full(6 * len(fac), L.WIRE_WEAK, dtype=np.int32)


def mean_field(edge_var, edge_fac, edge_role, prior_factors=()) -> Triples:
    """MeanFieldResolver (:599-621): marginals are flat products of all incoming messages (intermediate dependencies), messages depend weakly
    on marginals.  prior_factors: ids of opaque unary factors (their messages are set by the caller; they only appear in the marginals)."""
    edge_var, edge_fac, edge_role = np.asarray(edge_var, np.int64), np.asarray(edge_fac, np.int64), np.asarray(edge_role, np.int64)
    is_prior = np.isin(edge_fac, np.asarray(list(prior_factors), dtype=np.int64))
    fac, out, inn, prec = _by_factor(edge_var[~is_prior], edge_fac[~is_prior], edge_role[~is_prior])
    s1, d1, f1 = _mean_field_factors(fac, out, inn, prec)
    order = np.lexsort((edge_fac, edge_var))            # per variable its factors in ascending id order
    s2 = _rows(L.ITEM_INDIVIDUAL_MARGINAL, edge_var[order], 0)
    d2 = _rows(L.ITEM_MESSAGE_TO_VARIABLE, edge_var[order], edge_fac[order])
    f2 = np.full(len(order), L.WIRE_INTERMEDIATE, dtype=np.int32)
    return Triples(np.concatenate([s1, s2]), np.concatenate([d1, d2]), np.concatenate([f1, f2]))


def structured(edge_var, edge_fac, edge_role, clustered_factors, prior_factors=()) -> Triples:
    """StructuredResolver (:810-897): the factors named in clustered_factors treat their two Normal variables as ONE cluster — a JointMarginal
    that depends weakly on the two MessageToFactor signals and on the precision's marginal, linked to both variables; the messages to the two
    Normal variables depend on each other's MessageToFactor (strongly: belief propagation through the factor) and weakly on the precision's
    marginal; the message to the precision depends weakly on the joint.  All other NORMAL_PRECISION factors are wired mean-field (the
    reference's :likelihood branch).  The variable side is the default resolver's (CX_WIRE_DEFAULT_VARIABLE for every variable, ascending)."""
    edge_var, edge_fac, edge_role = np.asarray(edge_var, np.int64), np.asarray(edge_fac, np.int64), np.asarray(edge_role, np.int64)
    is_prior = np.isin(edge_fac, np.asarray(list(prior_factors), dtype=np.int64))
    fac, out, inn, prec = _by_factor(edge_var[~is_prior], edge_fac[~is_prior], edge_role[~is_prior])
    cl = np.isin(fac, np.asarray(clustered_factors, dtype=np.int64))
    s1, d1, f1 = _mean_field_factors(fac[~cl], out[~cl], inn[~cl], prec[~cl])
    f, a, b, g = fac[cl], np.minimum(out[cl], inn[cl]), np.maximum(out[cl], inn[cl]), prec[cl]      # the cluster in ascending id order
    joint = _rows(L.ITEM_JOINT_MARGINAL, np.zeros(len(f), np.int64), f)
    W, Z = L.WIRE_WEAK, 0
    # link_signal_to_variable! (:860): a variable's linked signals in the order of the calls, i.e. by ascending factor id — factor by factor,
    # member by member
    link_sig = np.repeat(joint, 2, axis=0)
    link_dep = _rows(L.ITEM_INDIVIDUAL_MARGINAL, np.stack([a, b], axis=1).reshape(-1), 0)
    parts = [
        # the joint: a weak dependency on each member's MessageToFactor (:858-868), then on the other cluster's marginal (:872-876)
        (joint, _rows(L.ITEM_MESSAGE_TO_FACTOR, a, f), W), (joint, _rows(L.ITEM_MESSAGE_TO_FACTOR, b, f), W),
        (joint, _rows(L.ITEM_INDIVIDUAL_MARGINAL, g, 0), W),
        # messages inside the cluster: strong dependency on the other member's MessageToFactor (:879-887) ...
        (_rows(L.ITEM_MESSAGE_TO_VARIABLE, a, f), _rows(L.ITEM_MESSAGE_TO_FACTOR, b, f), Z),
        (_rows(L.ITEM_MESSAGE_TO_VARIABLE, b, f), _rows(L.ITEM_MESSAGE_TO_FACTOR, a, f), Z),
        # ... and weak ones on the other clusters' marginals (:889-895)
        (_rows(L.ITEM_MESSAGE_TO_VARIABLE, a, f), _rows(L.ITEM_INDIVIDUAL_MARGINAL, g, 0), W),
        (_rows(L.ITEM_MESSAGE_TO_VARIABLE, b, f), _rows(L.ITEM_INDIVIDUAL_MARGINAL, g, 0), W),
        (_rows(L.ITEM_MESSAGE_TO_VARIABLE, g, f), joint, W),
    ]
    s2 = np.concatenate([link_sig] + [p[0] for p in parts]); d2 = np.concatenate([link_dep] + [p[1] for p in parts])
    f2 = np.concatenate([np.full(2 * len(f), L.WIRE_LINK, dtype=np.int32)] + [np.full(len(f), p[2], dtype=np.int32) for p in parts])
    vars_ = np.unique(edge_var)
    s3 = _rows(L.ITEM_INDIVIDUAL_MARGINAL, vars_, 0)
    return Triples(np.concatenate([s1, s2, s3]), np.concatenate([d1, d2, s3]),
                   np.concatenate([f1, f2, np.full(len(vars_), L.WIRE_DEFAULT_VARIABLE, dtype=np.int32)]))


def from_engine(engine) -> Triples:
    """The wiring of a constructed InferenceEngine (host mirror of the reference's API) as cx_graph_wire triples: whatever
    AbstractDependencyResolver ran over the engine (/root/reference/src/dependencies.jl:5-15), its add_dependency! calls are read back
    from the signals — per signal the dependencies in their order with their weak / intermediate bits (SignalDependenciesProps) and the
    listen bit (the dependency's listenmask), per variable the linked signals in link order.  A variable whose signals hang off
    ProductOfMessages nodes was wired by the default resolver's segment tree (those nodes cannot be named from outside): it is exported as
    CX_WIRE_DEFAULT_VARIABLE.  Factor side first, so that the default variable wiring finds the listeners it asks for."""
    from .inference_signal import InferenceSignalVariants as V
    from .model_engine import get_variable_linked_signals, get_variable_marginal
    from .signal import IS_INTERMEDIATE, IS_WEAK

    sig, dep, flags = [], [], []

    def name(s):
        v = s.variant
        if isinstance(v, V.MessageToFactor):
            return (L.ITEM_MESSAGE_TO_FACTOR, v.variable_id, v.factor_id)
        if isinstance(v, V.MessageToVariable):
            return (L.ITEM_MESSAGE_TO_VARIABLE, v.variable_id, v.factor_id)
        if isinstance(v, V.IndividualMarginal):
            return (L.ITEM_INDIVIDUAL_MARGINAL, v.variable_id, 0)
        if isinstance(v, V.JointMarginal):
            return (L.ITEM_JOINT_MARGINAL, 0, v.factor_id)
        raise TypeError(f"cx_graph_wire cannot name a signal of variant {v!r}")

    joints, seen_joint = [], set()

    def export(s):
        me = name(s)
        for i, d in enumerate(s.dependencies):
            if isinstance(d.variant, V.JointMarginal) and id(d) not in seen_joint:
                seen_joint.add(id(d)); joints.append(d)
            fl = (L.WIRE_WEAK if s.dependencies_props.get(i, IS_WEAK) else 0) | (L.WIRE_INTERMEDIATE if s.dependencies_props.get(i, IS_INTERMEDIATE) else 0)
            k = next(j for j, l in enumerate(d.listeners) if l is s)      # add_dependency! pushes listener and listenmask together
            if not d.listenmask[k]:
                fl |= L.WIRE_NO_LISTEN
            sig.append(me); dep.append(name(d)); flags.append(fl)

    var_ids = list(engine.get_variable_ids())
    for f in engine.get_factor_ids():
        for v in engine.get_connected_variable_ids(f):
            export(engine.get_connection_message_to_variable(v, f))
    for v in var_ids:      # joint marginals reachable only through a link
        for ls in get_variable_linked_signals(engine.get_variable(v)):
            if isinstance(ls.variant, V.JointMarginal) and id(ls) not in seen_joint:
                seen_joint.add(id(ls)); joints.append(ls)
    done = 0
    while done < len(joints):      # (a joint may depend on another joint: the structured resolver wires clusters to each other)
        export(joints[done]); done += 1
    for v in var_ids:
        variable = engine.get_variable(v)
        marg = get_variable_marginal(variable)
        own = [marg] + [engine.get_connection_message_to_factor(v, f) for f in engine.get_connected_factor_ids(v)]
        if any(isinstance(d.variant, V.ProductOfMessages) for s in own for d in s.dependencies):
            sig.append(name(marg)); dep.append(name(marg)); flags.append(L.WIRE_DEFAULT_VARIABLE)
        else:
            for s in own:
                export(s)
        for ls in get_variable_linked_signals(variable):
            sig.append(name(ls)); dep.append(name(marg)); flags.append(L.WIRE_LINK)
    while done < len(joints):
        export(joints[done]); done += 1
    return Triples(np.asarray(sig, dtype=np.int64).reshape(-1, 3), np.asarray(dep, dtype=np.int64).reshape(-1, 3), np.asarray(flags, dtype=np.int32))
