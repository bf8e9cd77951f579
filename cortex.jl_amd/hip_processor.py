"""HipProcessor — the AbstractInferenceRequestProcessor that puts the MI355X sweep behind the reference's plugin API
(src/inference_engine.jl:331-509).  The host keeps Signals, readiness bits and the scheduler; message values live in
HBM behind a cx_handle and the rule calls become launches through the C ABI (include/cortex_hip.h).

Modes:
  "per_signal"  every `process!` is a 1-element cx_update_batch followed by set_value!: the reference's exact
                execution order, one launch per message (schedule parity; slow by construction).
  "wavefront"   update_marginals! takes the call over: scan the currently pending signals
                (scan_inference_request, inference_engine.jl:540-546), compute the whole wavefront in ONE launch,
                set_value! them in scan order, repeat.  Same results on trees, O(depth) launches.  A request whose
                dependency graph has a cycle (a loopy factor graph) is handed back to the host scheduler and runs per
                signal: a wavefront there would be a Jacobi step the reference never takes.
  "sweep"       update_marginals! runs `n_sweeps` passes of the device schedule over the whole graph (cx_sweep) and
                marks the requested marginals computed.  The benchmarked path.
  "reference"   update_marginals!(engine, ids) is ONE cx_sweep_for(ids) under CX_SCHED_REFERENCE: the library replays what the
                reference's scheduler would do for exactly this request — the same signals in the same order, each from the values
                the reference's rule call would read, loops included — as one graph launch; plans are kept per readiness state, so
                a repeated call (new data, same request) costs one launch.  The readiness bits live in the library's shadow
                (csrc/cx_refsched.h); the host marks the requested marginals computed.  Scalar messages.
                With a USER resolver (anything but DefaultDependencyResolver) the engine's signals carry that resolver's
                add_dependency! calls when the processor is attached: they are read back (wiring.from_engine) and become the device's
                wiring (cx_graph_wire) — weak / intermediate / listen flags, joint marginals and linked signals included; factors with a
                NormalPrecisionFactor functional form get the variational rules their dependency lists select, marginals are then
                settable (set_value on a marginal signal: Gamma, NormalMeanPrecision, NormalMeanVariance, a number = observed).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Tuple

import struct
import numpy as np

from . import _lib as L
from .device import DeviceGraph
from .inference_engine import (AbstractInferenceRequestProcessor, InferenceEngine, request_inference_for,
                               scan_inference_request)
from .dependencies import DefaultDependencyResolver
from .inference_signal import InferenceSignalVariants as V
from .model_engine import get_factor_functional_form, get_variable_marginal
from .signal import ALL_FRESH, _M64, Signal, is_pending, set_value as _host_set_value


@dataclass(frozen=True)
class NormalMeanVariance:
    """The reference's test struct (test/runtests.jl:31-34)."""
    mean: float
    variance: float


@dataclass(frozen=True)
class GaussianAdditive:
    """functional_form of a 2-edge factor x_b = x_a + N(0, variance): the likelihood / transition of
    test/inference_engine_tests.jl:415-432 (variance 1.0 there)."""
    variance: float = 1.0


@dataclass(frozen=True)
class GaussianLinear:
    """x_out = a * x_in + b + N(0, variance); edges labelled :in / :out (Connection.label, model_engine.jl:182)."""
    a: float
    b: float
    variance: float


@dataclass(frozen=True, eq=False)
class MvNormalMeanCovariance:
    """d-dimensional Gaussian in moment form — what a device-computed message or marginal of a dim > 1 processor reads as."""
    mean: Any
    covariance: Any


@dataclass(frozen=True, eq=False)
class MvNormalMeanPrecision:
    """The reference's test struct (test/runtests.jl:70-77, `MvNormalMeanPrecision(mean, precision)`); accepted by set_value."""
    mean: Any
    precision: Any


@dataclass(frozen=True, eq=False)
class MvGaussianLinear:
    """functional_form of a 2-edge factor x_out = A x_in + N(0, Q) between d-dimensional variables; edges labelled :in / :out
    (Connection.label, model_engine.jl:182).  Factors that share one (A, Q) object share one parameter set on the device."""
    A: Any
    Q: Any


@dataclass(frozen=True)
class Beta:
    """The reference's test struct of the Beta-Bernoulli model (test/runtests.jl; `Beta(a, b)`).  On the device a Beta
    travels as its natural parameters (a - 1, b - 1) in the generic 2-parameter family, so the reference's product
    Beta(a + a' - 1, b + b' - 1) (test/inference_engine_tests.jl:270-277) is a plain sum."""
    a: float
    b: float


def default_factor_rule(factor) -> Tuple[int, Tuple[float, ...]]:
    ff = get_factor_functional_form(factor)
    if isinstance(ff, str) and ff.lstrip(":") == "bernoulli":     # functional_form = :bernoulli, test/...:323
        return L.FACTOR_BERNOULLI, ()
    if isinstance(ff, GaussianAdditive):
        return L.FACTOR_GAUSS_ADDITIVE, (ff.variance,)
    if isinstance(ff, GaussianLinear):
        return L.FACTOR_GAUSS_LINEAR, (ff.variance, ff.a, ff.b)
    if isinstance(ff, MvGaussianLinear):
        return L.FACTOR_GAUSS_LINEAR, ff          # dim > 1: attach() turns the (A, Q) object into a parameter-set index
    if type(ff).__name__ == "NormalPrecisionFactor":      # (defined below) out ~ N(mean, 1 / precision): variational rules, mode "reference" + a user resolver
        return L.FACTOR_NORMAL_PRECISION, ()
    return L.FACTOR_OPAQUE, ()


def _unique(signals):
    """first occurrences, in order (a signal may be reachable from several marginals)"""
    seen, uniq = set(), []
    for s in signals:
        if id(s) not in seen:
            seen.add(id(s)); uniq.append(s)
    return uniq


def request_scope(request):
    """ids of the signals a scan of this request can ever visit: every dependency of a visited signal, visiting (as
    process_dependencies! does, signal.jl:466-490) only through dependencies flagged intermediate"""
    from .signal import IS_INTERMEDIATE

    scope, visited, stack = set(), set(), list(request.marginals)
    while stack:
        node = stack.pop()
        props = node.dependencies_props
        for i, dep in enumerate(node.dependencies):
            scope.add(id(dep))
            if props.get(i, IS_INTERMEDIATE) and id(dep) not in visited:
                visited.add(id(dep)); stack.append(dep)
    return scope


def request_has_cycle(request) -> bool:
    """Does the dependency graph a request can reach hold a directed cycle?  Followed through EVERY dependency (a pending signal's
    inputs are recomputed by the scheduler whether or not the edge is flagged intermediate, once they become pending themselves).
    With the default resolver's wiring (dependencies.jl:17-126) this is exactly "the factor graph has a loop": a variable→factor
    message depends on the other factor→variable messages of its variable, those on the other variable→factor messages of their
    factor, and so on around the loop.  Iterative three-colour depth-first search, O(signals + dependencies)."""
    colour = {}
    for root in request.marginals:
        if id(root) in colour:
            continue
        colour[id(root)] = 1
        stack = [(root, 0)]
        while stack:
            node, i = stack[-1]
            deps = node.dependencies
            if i == len(deps):
                colour[id(node)] = 2
                stack.pop()
                continue
            stack[-1] = (node, i + 1)
            d = deps[i]
            c = colour.get(id(d), 0)
            if c == 1:
                return True
            if c == 0:
                colour[id(d)] = 1
                stack.append((d, 0))
    return False


def run_wavefronts(request, launch, stats: Optional[dict] = None):
    """The batched mode between per-signal dispatch and the whole-call takeover: `launch(front)` computes one set of mutually
    independent pending signals and set_value!s them (one device launch), repeated until nothing is pending, then the requested
    marginals.

    ONLY for requests whose dependency graph is acyclic (trees; `request_has_cycle`).  There a pending signal's inputs are all fresh,
    i.e. final for this call, so computing a whole frontier at once leaves what the reference's one-signal-at-a-time order leaves.
    On a graph with a loop every signal of a frontier would read the values its neighbours held BEFORE the frontier — a Jacobi step —
    where the reference's sequential pass (inference_engine.jl:575-608) reads the newest ones: a result the reference never computes.
    HipProcessor.update_marginals therefore checks the request first and hands a cyclic one back to the host scheduler, which runs
    it per signal in the reference's order (tests/test_gpu_loopy_plugin.py).

    The first frontier comes from a full scan of the request (scan_inference_request, inference_engine.jl:540-546: O(request)).
    After that a signal can only BECOME pending because one of its dependencies was just set — set_value! notifies exactly its
    listeners (signal.jl:232-253) — so the next frontier is looked for among the listeners of the signals of this one:
    O(frontier) per wavefront instead of O(request) (a chain of T states is 2T wavefronts: O(T²) host work with a scan each).
    When that set runs dry a full scan confirms that nothing is pending any more.  On trees the frontiers are those of the
    repeated full scan; stats (if given) counts wavefronts and full scans."""
    scope = request_scope(request)
    marginal_ids = {id(m) for m in request.marginals}
    front = _unique(scan_inference_request(request))
    deferred: List[Signal] = []
    n_fronts, n_scans = 0, 1
    while True:
        if not front:
            front = _unique(scan_inference_request(request))
            n_scans += 1
            if not front:
                break
        launch(front)
        n_fronts += 1
        cand = _unique(deferred + [l for s in front for on, l in zip(s.listenmask, s.listeners) if on])
        pend = [l for l in cand if id(l) in scope and id(l) not in marginal_ids and is_pending(l)]
        # a batch holds mutually independent signals: a pending signal that is a direct dependency of another pending one waits a
        # round (the scan hides it behind its listener in the same way, signal.jl:476-482)
        pid = {id(l) for l in pend}
        hidden = {id(d) for l in pend for d in l.dependencies if id(d) in pid}
        front = [l for l in pend if id(l) not in hidden]
        deferred = [l for l in pend if id(l) in hidden]
    final = [m for m in request.marginals if is_pending(m)]
    if final:
        launch(final)
        n_fronts += 1
    if stats is not None:
        stats["wavefronts"], stats["full_scans"] = n_fronts, n_scans


class HipValue:
    """What a device-computed Signal holds on the host: a handle, not the payload (`Signal.value` only has to differ
    from UndefValue() for is_computed, signal.jl:162-164).  `.mean` / `.variance` read the device on first use."""
    __slots__ = ("_proc", "_variant", "_cache")

    def __init__(self, proc, variant):
        self._proc, self._variant, self._cache = proc, variant, None

    def _fetch(self):
        if self._cache is None:
            self._cache = self._proc.read(self._variant)
        return self._cache

    @property
    def mean(self):
        return self._fetch().mean

    @property
    def variance(self):
        return self._fetch().variance

    @property
    def covariance(self):
        return self._fetch().covariance

    @property
    def a(self):
        return self._fetch().a

    @property
    def b(self):
        return self._fetch().b

    def __repr__(self):
        return f"HipValue({self._variant!r})"


class HipProcessor(AbstractInferenceRequestProcessor):
    def __init__(self, *, mode: str = "sweep", n_sweeps: int = 1, device: int = 0, schedule: int = L.SCHED_FUSED,
                 factor_rule: Callable = default_factor_rule, family: str = "gaussian", dim: int = 1):
        """dim: the dimension of every variable (1: scalars; 2, 3, 4: small matrices in registers; 64: the MFMA path).  With
        dim > 1 the factors are MvGaussianLinear, data are length-d arrays and messages MvNormalMeanCovariance."""
        if mode not in ("per_signal", "wavefront", "sweep", "reference"):
            raise ValueError(f"unknown mode {mode!r}")
        if mode == "reference":
            schedule = L.SCHED_REFERENCE
        if family not in ("gaussian", "beta"):
            raise ValueError(f"unknown family {family!r}")
        if dim != 1 and family != "gaussian":
            raise ValueError("dim > 1 is Gaussian")
        self.mode, self.n_sweeps, self.factor_rule, self.family, self.dim = mode, n_sweeps, factor_rule, family, dim
        # raises without a GPU: no CPU fallback
        self.dev = DeviceGraph(device=device, dim=dim, schedule=schedule,
                               family=L.FAMILY_GAUSSIAN if family == "gaussian" else L.FAMILY_NATURAL2)
        self.engine: Optional[InferenceEngine] = None
        self.launches = 0
        self._records: dict = {}
        self._record_owners: list = []
        self.execution_log: List[Any] = []   # variants in execution order (schedule-parity checks)
        self._cycle_key, self._cycle = None, False
        self.cyclic_requests = 0             # wavefront mode: calls handed back to the host scheduler because the request's graph has a loop

    # ---- build hook: flatten the bipartite graph through the reference's 7 accessors -------------------------
    def attach(self, engine: InferenceEngine):
        self.engine = engine
        ev, ef, role, fids, kinds, params = [], [], [], [], [], []
        psets = {}                             # id(functional_form) -> parameter-set index (dim > 1)
        self.gamma_variables = set()           # precisions of NormalPrecisionFactor factors: their marginals are Gamma(shape, scale)
        for f in engine.get_factor_ids():
            kind, p = self.factor_rule(engine.get_factor(f))
            if kind == L.FACTOR_NORMAL_PRECISION:
                if self.mode != "reference":
                    raise NotImplementedError("a NormalPrecisionFactor has variational rules only: HipProcessor(mode='reference') with a resolver that wires them, or HipVmpProcessor")
                fids.append(f); kinds.append(kind); params.append((0.0,) * L.NPARAM)
                for v, r in _normal_precision_roles(engine, f):
                    ev.append(v); ef.append(f); role.append(r)
                    if r == L.ROLE_PRECISION:
                        self.gamma_variables.add(v)
                continue
            if isinstance(p, MvGaussianLinear):
                if self.dim == 1:
                    raise TypeError("MvGaussianLinear factors need HipProcessor(dim=d)")
                if id(p) not in psets:
                    psets[id(p)] = len(psets)
                    self.dev.set_factor_matrices(psets[id(p)], np.asarray(p.A, dtype=np.float64), np.asarray(p.Q, dtype=np.float64))
                p = (float(psets[id(p)]),)
            elif self.dim > 1 and kind != L.FACTOR_OPAQUE:
                raise TypeError(f"HipProcessor(dim={self.dim}): factor {f} needs an MvGaussianLinear functional form")
            fids.append(f); kinds.append(kind); params.append(tuple(p) + (0.0,) * (L.NPARAM - len(p)))
            for v in engine.get_connected_variable_ids(f):
                ev.append(v); ef.append(f)
                role.append(L.ROLE_IN if str(engine.get_connection(v, f).label).lstrip(":") == "in" else L.ROLE_OUT)
        if not ev:
            return
        self.dev.graph_create(ev, ef, fids, kinds, np.asarray(params, dtype=np.float64), edge_role=role)
        if self.mode == "reference" and type(engine.dependency_resolver) is not DefaultDependencyResolver:
            # a user resolver ran over the engine (dependencies.jl:1-15): its add_dependency! calls, read back from the signals, become the
            # device's wiring (cx_graph_wire) — the device then schedules exactly what the host engine would
            from . import wiring
            t = wiring.from_engine(engine)
            self.dev.graph_wire(t.signals, t.dependencies, t.flags)

    # ---- data injection: set_value! on a message signal, mirrored to the device ---------------------------------
    def set_value(self, signal: Signal, value):
        variant = signal.variant
        if isinstance(variant, V.MessageToFactor):
            direction = L.TO_FACTOR
        elif isinstance(variant, V.MessageToVariable):
            direction = L.TO_VARIABLE
        elif isinstance(variant, V.IndividualMarginal) and self.mode == "reference" and self.dim == 1:
            # set_value!(get_variable_marginal(...), value): the initial q's and the data of a wiring whose messages depend on marginals
            vid = variant.variable_id
            if isinstance(value, Gamma):
                self.dev.set_marginals([vid], L.FORM_GAMMA, [value.shape, value.scale])
            elif isinstance(value, NormalMeanPrecision):
                self.dev.set_marginals([vid], L.FORM_MEAN_PRECISION, [value.mean, value.precision])
            elif isinstance(value, NormalMeanVariance):
                self.dev.set_marginals([vid], L.FORM_MOMENT, [value.mean, value.variance])
            elif isinstance(value, (bool, int, float, np.floating)):
                self.dev.set_marginals([vid], L.FORM_POINT, [float(value)])
            else:
                raise TypeError(f"HipProcessor.set_value: no device form for a marginal of type {type(value).__name__}")
            _host_set_value(signal, value)
            return
        else:
            raise TypeError("HipProcessor.set_value: message signals carry device payloads (marginals too in mode 'reference')")
        if self.dim > 1:
            d = self.dim
            if isinstance(value, MvNormalMeanCovariance):
                form, payload = L.FORM_MOMENT, np.concatenate([np.asarray(value.mean, dtype=np.float64).reshape(d), np.asarray(value.covariance, dtype=np.float64).reshape(d * d)])
            elif isinstance(value, MvNormalMeanPrecision):
                W = np.asarray(value.precision, dtype=np.float64).reshape(d, d)
                form, payload = L.FORM_NATURAL, np.concatenate([W @ np.asarray(value.mean, dtype=np.float64).reshape(d), W.reshape(d * d)])
            else:                              # an observed datum: the `Real` branch of the reference's rule, d-dimensional
                form, payload = L.FORM_POINT, np.asarray(value, dtype=np.float64).reshape(d)
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, form, payload)
        elif isinstance(value, (bool, int, float, np.floating, np.bool_)):
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, L.FORM_POINT, [float(value)])
        elif isinstance(value, Beta):
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, L.FORM_NATURAL, [value.a - 1.0, value.b - 1.0])
        elif isinstance(value, Gamma):         # a message to a precision (a prior): natural pair (shape - 1, rate)
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, L.FORM_NATURAL, [value.shape - 1.0, 1.0 / value.scale])
        elif isinstance(value, NormalMeanPrecision):
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, L.FORM_NATURAL, [value.mean * value.precision, value.precision])
        else:
            self.dev.set_messages([variant.variable_id], [variant.factor_id], direction, L.FORM_MOMENT,
                                  [float(value.mean), float(value.variance)])
        _host_set_value(signal, value)

    def read(self, variant):
        form = L.FORM_MOMENT if self.family == "gaussian" else L.FORM_NATURAL
        if isinstance(variant, V.JointMarginal):
            mean, cov = self.dev.get_joint_marginals([variant.factor_id])
            return mean[0], cov[0]
        if isinstance(variant, V.IndividualMarginal):
            m = self.dev.get_marginals([variant.variable_id])[0]
            if variant.variable_id in getattr(self, "gamma_variables", ()):
                return Gamma(float(m[0]), float(m[1]))
        elif isinstance(variant, V.MessageToVariable):
            m = self.dev.get_messages([variant.variable_id], [variant.factor_id], L.TO_VARIABLE, form)[0]
        elif isinstance(variant, V.MessageToFactor):
            m = self.dev.get_messages([variant.variable_id], [variant.factor_id], L.TO_FACTOR, form)[0]
        elif isinstance(variant, V.ProductOfMessages):
            m = self.dev.get_products([variant.variable_id], [variant.range[0]], [variant.range[1]], form)[0]
        else:
            raise TypeError(f"no device payload for {variant!r}")
        if self.dim > 1:
            d = self.dim
            return MvNormalMeanCovariance(np.array(m[:d]), np.array(m[d:]).reshape(d, d))
        if self.family == "beta":
            return Beta(float(m[0]) + 1.0, float(m[1]) + 1.0)
        return NormalMeanVariance(float(m[0]), float(m[1]))

    # ---- rules: each is a 1-element batch (the scalar fallback of SURVEY §8b) -----------------------------------
    @staticmethod
    def _item(variant):
        if isinstance(variant, V.MessageToVariable):
            return L.ITEM_MESSAGE_TO_VARIABLE, variant.variable_id, variant.factor_id
        if isinstance(variant, V.MessageToFactor):
            return L.ITEM_MESSAGE_TO_FACTOR, variant.variable_id, variant.factor_id
        if isinstance(variant, V.IndividualMarginal):
            return L.ITEM_INDIVIDUAL_MARGINAL, variant.variable_id, 0
        if isinstance(variant, V.ProductOfMessages):      # 1-based inclusive range, as in Julia (inference_signal.jl:62-66)
            # the reference's range indexes `factors_connected_to_variable` (dependencies.jl:128-173), i.e. whatever order the model
            # engine's get_connected_factor_ids returns; the device resolves it over ASCENDING factor ids (cortex_hip.h).  An engine
            # that lists factors in another order would multiply the wrong subsets without any error: refuse it here.
            fs = tuple(variant.factors_connected_to_variable)
            if any(a >= b for a, b in zip(fs, fs[1:])):
                raise ValueError(f"ProductOfMessages of variable {variant.variable_id}: factors_connected_to_variable is not in ascending id order "
                                 "(the device resolves ranges over ascending factor ids)")
            return L.ITEM_PRODUCT_OF_MESSAGES, variant.variable_id, L.item_range(variant.range[0], variant.range[1])
        if isinstance(variant, V.JointMarginal):
            return L.ITEM_JOINT_MARGINAL, 0, variant.factor_id
        raise NotImplementedError(f"The HIP processor has no rule for {type(variant).__name__}")

    def _launch(self, variants):
        # every signal's record (cx_item: kind, 0, variable_id, factor_id) is packed once, at its first launch: a chain of T states is
        # 2T wavefronts of three signals, and building ctypes items per launch cost as much as the launch itself
        recs = self._records
        try:
            buf = b"".join([recs[id(v)] for v in variants])
        except KeyError:
            for v in variants:
                if id(v) not in recs:
                    k, var, fac = self._item(v)
                    recs[id(v)] = struct.pack("<iiqq", int(k), 0, int(var), int(fac))
                    self._record_owners.append(v)            # keeps id(v) unique for the processor's lifetime
            buf = b"".join([recs[id(v)] for v in variants])
        self.dev.update_batch_packed(buf, len(variants))
        self.launches += 1
        self.execution_log.extend(variants)

    def compute_message_to_variable(self, engine, variant, signal, dependencies):
        self._launch([variant])
        return HipValue(self, variant)

    compute_message_to_factor = compute_message_to_variable
    compute_individual_marginal = compute_message_to_variable
    compute_product_of_messages = compute_message_to_variable     # inference_engine.jl:439-449
    compute_joint_marginal = compute_message_to_variable          # inference_engine.jl:469-477

    # ---- whole-call takeover (a Julia method of update_marginals! specialised on the processor type) -------------
    def update_marginals(self, engine, ids) -> bool:
        if self.mode == "per_signal":
            return False                       # the generic scheduler drives process! one signal at a time
        if self.mode == "reference":
            # ONE update_marginals! of the reference, replayed on the device (also computes the requested marginals).  A repeated request
            # (the iteration a user writes around the call) keeps its id array and its marginal signals: what is left on the host per call
            # is one set_value! per requested marginal
            key = (id(engine), len(ids), ids[0] if len(ids) else None, ids[-1] if len(ids) else None)
            kept = getattr(self, "_ref_request", None)
            if kept is None or kept[0] != key or kept[1] != list(ids):
                margs = [get_variable_marginal(engine.get_variable(vid)) for vid in ids]
                kept = (key, list(ids), np.ascontiguousarray(ids, dtype=np.int64), margs, [HipValue(self, m.variant) for m in margs])
                self._ref_request = kept
            self.dev.sweep_for(kept[2])
            self.launches += 1
            # set_value!(marginal, handle) per requested marginal (signal.jl:232-253), spelt out: the handle object is kept and only
            # forgets what it read last time; a marginal that somebody listens to goes through the general function
            fresh_mask = ~ALL_FRESH & _M64
            for m, hv in zip(kept[3], kept[4]):
                if m.listeners:
                    hv._cache = None
                    _host_set_value(m, hv)
                    continue
                hv._cache = None
                m.value = hv
                ch = m.dependencies_props.chunks
                for c in range(len(ch)):
                    ch[c] &= fresh_mask
                m.is_potentially_pending = False
                m._is_pending = False
            return True
        if self.mode == "sweep":
            self.dev.sweep(self.n_sweeps)
            self.refresh_marginals(ids)        # marginals of the messages the last sweep produced
            for vid in ids:
                m = get_variable_marginal(engine.get_variable(vid))
                _host_set_value(m, HipValue(self, m.variant))
            return True
        def launch(front):
            self._launch([s.variant for s in front])
            for s in front:
                _host_set_value(s, HipValue(self, s.variant))

        request = request_inference_for(engine, ids)
        key = tuple(ids)
        if self._cycle_key != key:                 # the wiring is fixed at construction: one search per distinct request
            self._cycle_key, self._cycle = key, request_has_cycle(request)
        if self._cycle:
            # a loopy request: the generic scheduler runs it, one process! per signal in the reference's order (request_inference_for
            # only flags potentially-pending signals, which the scheduler's own call repeats: handing back is free of side effects)
            self.cyclic_requests += 1
            return False
        run_wavefronts(request, launch)
        return True

    def refresh_marginals(self, ids):
        self.dev.update_batch([L.ITEM_INDIVIDUAL_MARGINAL] * len(ids), list(ids), [0] * len(ids))


# ---- variational families --------------------------------------------------------------------------
@dataclass(frozen=True)
class NormalMeanPrecision:
    """test/runtests.jl:48-56"""
    mean: float
    precision: float


@dataclass(frozen=True)
class Gamma:
    """test/runtests.jl:60-67"""
    shape: float
    scale: float


@dataclass(frozen=True)
class NormalPrecisionFactor:
    """functional_form of a 3-edge factor out ~ N(mean, 1 / precision): the :likelihood and :transition factors of
    test/inference_engine_tests.jl:691-715.  `roles` maps a connected variable's *name* to its role
    ("out" | "mean" | "precision"); with two variables of one name (x_i, x_{i+1}) the lower id is the mean."""
    roles: Tuple[Tuple[str, str], ...]


def _normal_precision_roles(engine, f):
    """(variable id, CX_ROLE_*) of a NormalPrecisionFactor's three variables, ascending ids; with two variables of one name the lower id is the mean"""
    roles = dict(get_factor_functional_form(engine.get_factor(f)).roles)
    seen_mean, out = False, []
    for v in engine.get_connected_variable_ids(f):
        r = roles[str(engine.get_variable(v).name).lstrip(":")]
        if r == "both":
            r = "out" if seen_mean else "mean"
            seen_mean = True
        out.append((v, {"out": L.ROLE_OUT, "mean": L.ROLE_IN, "precision": L.ROLE_PRECISION}[r]))
    return out


class HipVmpValue:
    """Host-side handle of a device-held marginal of the variational families (`.mean`, `.precision` / `.shape`,
    `.scale` read the device on first use)."""
    __slots__ = ("_proc", "_vid", "_cache")

    def __init__(self, proc, vid):
        self._proc, self._vid, self._cache = proc, vid, None

    def get(self):
        if self._cache is None:
            self._cache = self._proc.read_marginal(self._vid)
        return self._cache

    def __getattr__(self, name):
        return getattr(self.get(), name)

    def __repr__(self):
        return f"HipVmpValue(variable {self._vid})"


class HipVmpProcessor(AbstractInferenceRequestProcessor):
    """The variational families behind the reference's plugin API.  With weak dependencies every `update_marginals!`
    is a whole-call takeover (cx_update_marginals); initial marginals and data enter through `set_value` on the
    marginal signals, exactly where the reference's tests call `set_value!(get_variable_marginal(...), ...)`.

    family = "mean_field" (MeanFieldResolver wiring) or "structured" (StructuredResolver wiring); the engine is built
    with `resolve_dependencies=False`: the dependency wiring these resolvers would create is what the device path
    implements, the host keeps only the marginal signals."""

    def __init__(self, *, family: str = "structured", device: int = 0, schedule: int = L.SCHED_CHAIN_SCAN):
        if family not in ("mean_field", "structured"):
            raise ValueError(f"unknown family {family!r}")
        fam = L.FAMILY_VMP_MEAN_FIELD if family == "mean_field" else L.FAMILY_VMP_STRUCTURED
        self.dev = DeviceGraph(device=device, schedule=schedule, family=fam)
        self.engine: Optional[InferenceEngine] = None
        self.kind = {}

    def attach(self, engine: InferenceEngine):
        self.engine = engine
        ev, ef, role, fids = [], [], [], []
        for f in engine.get_factor_ids():
            ff = get_factor_functional_form(engine.get_factor(f))
            if not isinstance(ff, NormalPrecisionFactor):
                raise NotImplementedError(f"The HIP VMP processor has no rule for a factor with functional form {ff!r}")
            roles = dict(ff.roles)
            seen_mean = False
            fids.append(f)
            for v in engine.get_connected_variable_ids(f):          # ascending ids
                name = str(engine.get_variable(v).name).lstrip(":")
                r = roles[name]
                if r == "both":                                      # two variables of one name: lower id = mean
                    r = "out" if seen_mean else "mean"
                    seen_mean = True
                ev.append(v); ef.append(f)
                role.append({"out": L.ROLE_OUT, "mean": L.ROLE_IN, "precision": L.ROLE_PRECISION}[r])
                self.kind[v] = "gamma" if r == "precision" else "normal"
        self.dev.graph_create(ev, ef, fids, [L.FACTOR_NORMAL_PRECISION] * len(fids), np.zeros(len(fids)), edge_role=role)

    def set_value(self, signal: Signal, value):
        variant = signal.variant
        if not isinstance(variant, V.IndividualMarginal):
            raise TypeError("HipVmpProcessor.set_value: the state of the variational families is the set of marginals")
        vid = variant.variable_id
        if isinstance(value, Gamma):
            self.dev.set_marginals([vid], L.FORM_GAMMA, [value.shape, value.scale])
        elif isinstance(value, NormalMeanPrecision):
            self.dev.set_marginals([vid], L.FORM_MEAN_PRECISION, [value.mean, value.precision])
        elif isinstance(value, (int, float, np.floating)):
            self.dev.set_marginals([vid], L.FORM_POINT, [float(value)])
        else:
            raise TypeError(f"HipVmpProcessor.set_value: no device form for {type(value).__name__}")
        _host_set_value(signal, value)

    def read_marginal(self, vid):
        a, b = self.dev.get_marginals([vid])[0]
        return Gamma(float(a), float(b)) if self.kind[vid] == "gamma" else NormalMeanPrecision(float(a), float(b))

    def update_marginals(self, engine, ids) -> bool:
        self.dev.update_marginals(list(ids))
        for vid in ids:
            m = get_variable_marginal(engine.get_variable(vid))
            _host_set_value(m, HipVmpValue(self, vid))
        return True
