"""Pins the CPU checker (oracle/cortex_ref.c) to the reference's OWN known-answer tests.

Every test below is a transcription of the *assertions* (inputs and expected outputs — data, not code) of a
test item of the reference; the item is named in the docstring.  Julia is not installed in this container
and the reference cannot run here, so these known answers + the exact solvers of oracle/exact.py are what
the oracle's parity rests on ("oracle").
"""
import numpy as np
import pytest

from oracle import exact, ref


class MirrorSignals:
    """The same free-signal surface as ref.Engine, backed by the product's host-side mirror
    (cortex.jl_amd/signal.py): the reference's known answers pin BOTH restatements."""

    def __init__(self):
        import cortex.jl_amd as cx
        self.cx = cx

    def signal(self):
        return self.cx.Signal()

    def set_value(self, s, value):
        self.cx.set_value(s, value)

    def add_dependency(self, s, d, weak=False, listen=True, check_computed=True, intermediate=False):
        self.cx.add_dependency(s, d, weak=weak, listen=listen, check_computed=check_computed, intermediate=intermediate)

    def is_pending(self, s):
        return self.cx.is_pending(s)

    def is_computed(self, s):
        return self.cx.is_computed(s)

    def dependencies(self, s):
        return list(self.cx.get_dependencies(s))

    def listeners(self, s):
        return list(self.cx.get_listeners(s))

    def chunks(self, s):
        return list(s.dependencies_props.chunks)

    def process_dependencies(self, s, fn, retry=False):
        return self.cx.process_dependencies(fn, s, retry=retry)

    def compute(self, s, strategy, force=False, skip_if_no_listeners=False):
        self.cx.compute(strategy, s, force=force, skip_if_no_listeners=skip_if_no_listeners)

    def value(self, s):
        v = self.cx.get_value(s)
        return None if isinstance(v, self.cx.UndefValue) else v


@pytest.fixture(params=["oracle", "mirror"])
def E(request):
    return ref.Engine(ref.P_SSM_BP) if request.param == "oracle" else MirrorSignals()


def val(E, s):
    """get_value as a plain number, None for UndefValue() — on either restatement"""
    if hasattr(E, "value"):
        return E.value(s)
    tag, a, _ = E.get_value(s)
    return None if tag == ref.UNDEF else a


def S(E, value=None):
    """Signal() / Signal(value)  (signal.jl:107-114)"""
    s = E.signal()
    if value is not None:
        E.set_value(s, value)
    return s


# ------------------------------------------------------------------ test/signal_tests.jl

def test_basic_signal_operations(E):
    """signal_tests.jl:1-21 "Basic Signal Operations" (the untyped half: typed Signals are Julia's type system)."""
    s = S(E, 42)
    assert val(E, s) == 42
    E.set_value(s, 100)
    assert val(E, s) == 100


def test_empty_signal_creation(E):
    """signal_tests.jl:69-88."""
    s = S(E)
    assert val(E, s) is None                       # UndefValue()
    assert E.dependencies(s) == [] and E.listeners(s) == []
    assert not E.is_pending(s) and not E.is_computed(s)


def test_signal_creation_with_value_sets_computed(E):
    """signal_tests.jl:90-97."""
    s = S(E, 10)
    assert val(E, s) == 10 and E.is_computed(s) and not E.is_pending(s)


def test_signal_computed_with_a_strategy(E):
    """signal_tests.jl:834-871 "A Signal Can Be Computed With A Lambda Function", basic case."""
    s1, s2, s3 = S(E, 1), S(E, 2), S(E)
    E.add_dependency(s3, s1); E.add_dependency(s3, s2)
    assert E.is_pending(s3) and not E.is_computed(s3)
    strategy = lambda _, deps: sum(val(E, d) for d in deps)
    E.compute(s3, strategy)
    assert E.is_computed(s3) and not E.is_pending(s3) and val(E, s3) == 3
    with pytest.raises(ValueError):                # ArgumentError: no longer pending
        E.compute(s3, strategy)
    E.compute(s3, strategy, force=True)
    assert val(E, s3) == 3 and not E.is_pending(s3)
    E.set_value(s1, 10); E.set_value(s2, 20)
    assert E.is_pending(s3)
    E.compute(s3, strategy)
    assert E.is_computed(s3) and not E.is_pending(s3) and val(E, s3) == 30


def test_pyramid_of_signals(E):
    """signal_tests.jl:873-916."""
    s01, s02, s11, s12 = S(E, 1), S(E, 2), S(E, 3), S(E, 4)
    s21, s22, s3 = S(E), S(E), S(E)
    E.add_dependency(s21, s01); E.add_dependency(s21, s02)
    E.add_dependency(s22, s11); E.add_dependency(s22, s12)
    E.add_dependency(s3, s21); E.add_dependency(s3, s22)
    assert E.is_pending(s21) and E.is_pending(s22) and not E.is_computed(s21) and not E.is_computed(s22)
    assert not E.is_pending(s3) and not E.is_computed(s3)
    strategy = lambda _, deps: sum(val(E, d) for d in deps)
    E.compute(s21, strategy); E.compute(s22, strategy)
    assert not E.is_pending(s21) and not E.is_pending(s22)
    assert E.is_pending(s3) and not E.is_computed(s3)          # pending once its dependencies are computed
    E.compute(s3, strategy)
    assert E.is_computed(s3) and not E.is_pending(s3) and val(E, s3) == 10


def test_intermediate_dependencies_are_listed(E):
    """signal_tests.jl:919-931."""
    source, intermediate, derived = S(E), S(E), S(E)
    E.add_dependency(intermediate, source)
    E.add_dependency(derived, intermediate, intermediate=True)
    assert E.dependencies(derived) == [intermediate] and E.dependencies(intermediate) == [source]


@pytest.mark.parametrize("retry", [False, True])
def test_process_dependencies_reports_that_something_was_processed(E, retry):
    """signal_tests.jl:1061-1082."""
    source, intermediate, derived = S(E), S(E), S(E)
    E.add_dependency(intermediate, source)
    E.add_dependency(derived, intermediate, intermediate=True)
    attempted = []

    def f(dep):
        attempted.append(dep)
        return dep is source or dep == source
    assert E.process_dependencies(derived, f, retry=retry)
    assert len(attempted) >= 1


def test_compute_skips_a_signal_without_listeners_unless_told_otherwise(E):
    """signal_tests.jl:1116-1133."""
    s = S(E, 1)
    E.compute(s, lambda sig, deps: 2, skip_if_no_listeners=True)
    assert val(E, s) == 1
    seen = []

    def strategy(sig, deps):
        seen.append(len(deps))
        return 2
    E.compute(s, strategy, force=True, skip_if_no_listeners=False)
    assert seen == [0] and val(E, s) == 2


def test_add_dependency_basic(E):
    """signal_tests.jl:99-135 "Add Dependency Basic"."""
    a, b = S(E, 1), S(E, 2)
    assert E.dependencies(a) == [] and E.listeners(a) == [] and not E.is_pending(a) and not E.is_pending(b)
    E.add_dependency(a, b)
    assert E.dependencies(a) == [b] and E.listeners(a) == [] and E.dependencies(b) == [] and E.listeners(b) == [a]
    assert not E.is_pending(a) and not E.is_pending(b)
    E.set_value(b, 3)
    assert E.is_pending(a) and not E.is_pending(b)


def test_add_dependency_initialized(E):
    """signal_tests.jl:164-182."""
    dep, s = S(E, 1), S(E)
    assert not E.is_pending(s) and not E.is_computed(s) and E.is_computed(dep)
    E.add_dependency(s, dep)
    assert E.is_pending(s) and not E.is_computed(s)


def test_single_weak_dependency(E):
    """signal_tests.jl:184-222 (non-initialized and initialized weak dependency)."""
    s1, s2 = S(E), S(E)
    E.add_dependency(s2, s1, weak=True)
    assert not E.is_pending(s2)
    E.set_value(s1, 10)
    assert E.is_pending(s2) and not E.is_computed(s2)
    s1, s2 = S(E, 1), S(E)
    E.add_dependency(s2, s1, weak=True)
    assert E.is_pending(s2)
    E.set_value(s1, 10)
    assert E.is_pending(s2)


def test_dependency_without_check_computed(E):
    """signal_tests.jl:224-242."""
    s1, s2 = S(E, 1), S(E)
    E.add_dependency(s2, s1, check_computed=False)
    assert not E.is_pending(s2)
    E.set_value(s1, 10)
    assert E.is_pending(s2) and not E.is_computed(s2)


def test_many_strong_dependencies(E):
    """signal_tests.jl:244-285."""
    s1, s2, s3, d = S(E), S(E), S(E), S(E)
    for s in (s1, s2, s3):
        E.add_dependency(d, s)
    assert E.dependencies(d) == [s1, s2, s3] and E.listeners(s1) == [d]
    assert not E.is_pending(d)
    E.set_value(s1, 1)
    assert not E.is_pending(d)
    E.set_value(s2, 2)
    assert not E.is_pending(d)
    E.set_value(s3, 3)
    assert E.is_pending(d) and not E.is_computed(d)
    E.set_value(d, 10)
    assert not E.is_pending(d) and E.is_computed(d)


def test_update_marks_pending(E):
    """signal_tests.jl:287-331."""
    for init in (False, True):
        s1, s2 = (S(E, 1), S(E, 2)) if init else (S(E), S(E))
        E.add_dependency(s1, s2)
        assert not E.is_pending(s1) and not E.is_pending(s2)
        E.set_value(s2, 3)
        assert E.is_pending(s1) and not E.is_pending(s2) and E.is_computed(s2)
        assert E.is_computed(s1) == init


def test_weak_dependencies_basic(E):
    """signal_tests.jl:333-366."""
    weak, strong, d = S(E, 1), S(E, 2), S(E)
    E.add_dependency(d, weak, weak=True)
    E.add_dependency(d, strong)
    assert E.is_pending(d) and not E.is_computed(d)
    E.set_value(d, 10)
    assert not E.is_pending(d)
    E.set_value(strong, 3)
    assert E.is_pending(d)
    E.set_value(d, 11)
    assert not E.is_pending(d)
    E.set_value(weak, 4)
    assert not E.is_pending(d)
    E.set_value(strong, 5)
    assert E.is_pending(d)


def test_many_weak_dependencies(E):
    """signal_tests.jl:368-440."""
    w1, w2, st, d = S(E), S(E), S(E), S(E)
    E.add_dependency(d, w1, weak=True)
    E.add_dependency(d, w2, weak=True)
    E.add_dependency(d, st)
    assert not E.is_pending(d)
    E.set_value(st, 10)
    assert not E.is_pending(d)
    E.set_value(w1, 1)
    assert not E.is_pending(d)
    E.set_value(w2, 2)
    assert E.is_pending(d)
    E.set_value(d, 100)
    assert not E.is_pending(d) and E.is_computed(d)
    E.set_value(st, 11)
    assert E.is_pending(d)
    E.set_value(d, 101)
    assert not E.is_pending(d)
    E.set_value(w1, 3)
    assert not E.is_pending(d)
    E.set_value(st, 333)
    assert E.is_pending(d)


def test_duplicate_dependency_quirk(E):
    """signal_tests.jl:442-465: the second duplicate is never notified."""
    s1, s2 = S(E), S(E)
    E.add_dependency(s1, s2)
    E.add_dependency(s1, s2)
    assert E.dependencies(s1) == [s2, s2] and E.listeners(s2) == [s1, s1]
    E.set_value(s2, 1)
    assert not E.is_pending(s1)


def test_circular_dependencies(E):
    """signal_tests.jl:467-507."""
    s1, s2 = S(E), S(E)
    E.add_dependency(s1, s2)
    E.add_dependency(s2, s1)
    assert not E.is_pending(s1) and not E.is_pending(s2)
    E.set_value(s1, 1)
    assert not E.is_pending(s1) and E.is_pending(s2)
    E.set_value(s2, 2)
    assert E.is_pending(s1) and not E.is_pending(s2)
    E.set_value(s2, 3)
    assert E.is_pending(s1) and not E.is_pending(s2)
    E.set_value(s1, 4)
    assert not E.is_pending(s1) and E.is_pending(s2)


def test_self_dependency_is_noop(E):
    """signal_tests.jl:509-521."""
    s = S(E)
    E.add_dependency(s, s)
    assert E.dependencies(s) == [] and E.listeners(s) == [] and not E.is_pending(s)


def test_pending_state_logic_coverage(E):
    """signal_tests.jl:523-591."""
    d, st = S(E), S(E)
    E.add_dependency(d, st)
    assert not E.is_pending(d)
    E.set_value(d, 1)
    assert not E.is_pending(d)
    d, st = S(E), S(E)
    E.add_dependency(d, st)
    E.set_value(st, 10)
    assert E.is_pending(d)
    d, w = S(E), S(E)
    E.add_dependency(d, w, weak=True)
    assert not E.is_pending(d)
    E.set_value(d, 1)
    assert not E.is_pending(d)
    E.set_value(w, 10)
    assert E.is_pending(d)
    d, st = S(E, 1), S(E, 10)
    E.add_dependency(d, st)
    assert not E.is_pending(d)
    E.set_value(d, 100)
    assert not E.is_pending(d)
    E.set_value(st, 101)
    assert E.is_pending(d)
    E.set_value(d, 102)
    assert not E.is_pending(d)
    E.set_value(st, 103)
    assert E.is_pending(d)
    d, w, st = S(E), S(E), S(E)
    E.add_dependency(d, w, weak=True)
    E.add_dependency(d, st)
    assert not E.is_pending(d)
    E.set_value(w, 1)
    assert not E.is_pending(d)
    E.set_value(st, 2)
    assert E.is_pending(d)


def test_chain_of_signals(E):
    """signal_tests.jl:593-637."""
    s1, s2, s3 = S(E, 1), S(E), S(E)
    E.add_dependency(s2, s1)
    E.add_dependency(s3, s2)
    P = lambda: (E.is_pending(s1), E.is_pending(s2), E.is_pending(s3))
    assert P() == (False, True, False)
    E.set_value(s1, 2); assert P() == (False, True, False)
    E.set_value(s2, 3); assert P() == (False, False, True)
    E.set_value(s3, 4); assert P() == (False, False, False)
    E.set_value(s1, 5); assert P() == (False, True, False)
    E.set_value(s2, 6); assert P() == (False, False, True)
    E.set_value(s3, 7); assert P() == (False, False, False)


def test_not_listening_dependency(E):
    """signal_tests.jl:639-710."""
    s1, s2 = S(E, 1), S(E, 2)
    E.add_dependency(s2, s1, listen=False)
    assert not E.is_pending(s2)
    E.set_value(s1, 10)
    assert not E.is_pending(s2)
    s1, s2 = S(E, 1), S(E, 2)
    E.add_dependency(s2, s1, listen=False, weak=True)
    assert E.is_pending(s2)
    E.set_value(s1, 10)
    assert E.is_pending(s2)
    s1, s2 = S(E, 1), S(E, 2)
    E.add_dependency(s2, s1, listen=False, check_computed=False)
    assert not E.is_pending(s2)
    E.set_value(s1, 10)
    assert not E.is_pending(s2)
    s1, s2, s3 = S(E), S(E), S(E)
    E.add_dependency(s3, s1, listen=False)
    E.add_dependency(s3, s2)
    assert not E.is_pending(s3)
    E.set_value(s2, 10)
    assert not E.is_pending(s3)
    E.set_value(s1, 10)
    assert not E.is_pending(s3)
    E.set_value(s2, 30)
    assert E.is_pending(s3)


def test_computed_then_uncomputed_dependency(E):
    """signal_tests.jl:712-749."""
    s1, s2, d = S(E, 1), S(E), S(E)
    E.add_dependency(d, s1)
    assert E.is_pending(d)
    E.add_dependency(d, s2)
    assert not E.is_pending(d)
    s1, s2, d = S(E, 1), S(E), S(E)
    E.add_dependency(d, s1, check_computed=True)
    assert E.is_pending(d)
    E.add_dependency(d, s2, check_computed=False)
    assert E.is_pending(d)


def test_nibble_layout_and_many_dependencies(E):
    """signal.jl:17-45,507-526 / docs/src/signals.md:653-683: 4 bits per dependency (I=1, W=2, C=4, F=8), 16 per
    UInt64; exercises the multi-chunk path of is_meeting_pending_criteria (signal.jl:678-727)."""
    d = S(E)
    deps = [S(E) for _ in range(35)]
    for i, s in enumerate(deps):
        E.add_dependency(d, s, weak=(i % 3 == 0), intermediate=(i % 5 == 0))
    ch = E.chunks(d)
    assert len(ch) == 3
    for i in range(35):
        nib = (ch[i // 16] >> (4 * (i % 16))) & 0xF
        assert nib == (2 if i % 3 == 0 else 0) | (1 if i % 5 == 0 else 0)
    for i, s in enumerate(deps[:-1]):
        E.set_value(s, float(i))
        assert not E.is_pending(d)
    E.set_value(deps[-1], 1.0)
    assert E.is_pending(d)
    ch = E.chunks(d)
    for i in range(35):
        assert (ch[i // 16] >> (4 * (i % 16))) & 0xC == 0xC  # computed + fresh
    E.set_value(d, 0.0)
    ch = E.chunks(d)
    for i in range(35):
        assert (ch[i // 16] >> (4 * (i % 16))) & 0xC == 0x4  # fresh cleared by set_value! (signal.jl:241)
    assert not E.is_pending(d)
    # exactly 16 and 32 dependencies: the "<< 64" corner of signal.jl:716
    for n in (16, 32):
        d = S(E)
        deps = [S(E) for _ in range(n)]
        for s in deps:
            E.add_dependency(d, s)
        for s in deps:
            assert not E.is_pending(d)
            E.set_value(s, 1.0)
        assert E.is_pending(d)


def test_process_dependencies_visit_order(E):
    """signal_tests.jl:933-1029: six (retry, callback) cases on source -> intermediate -> derived."""
    source, inter, derived = S(E), S(E), S(E)
    E.add_dependency(inter, source)
    E.add_dependency(derived, inter, intermediate=True)

    def run(retry, fn):
        seen = []
        r = E.process_dependencies(derived, lambda s: (seen.append(s), fn(s))[1], retry=retry)
        return seen, r

    for retry in (False, True):
        seen, r = run(retry, lambda s: False)
        assert seen == [inter, source] and r is False
    for retry in (False, True):
        seen, r = run(retry, lambda s: True)
        assert seen == [inter] and r is True
    seen, r = run(False, lambda s: s != inter)
    assert seen == [inter, source] and r is True
    seen, r = run(True, lambda s: s != inter)
    assert seen == [inter, source, inter] and r is True


def test_process_dependencies_does_not_descend_non_intermediate(E):
    """signal_tests.jl:1031-1059."""
    source, mid, derived = S(E), S(E), S(E)
    E.add_dependency(mid, source)
    E.add_dependency(derived, mid)  # NOT intermediate
    seen = []
    r = E.process_dependencies(derived, lambda s: (seen.append(s), False)[1], retry=True)
    assert seen == [mid] and r is False


def test_compute_non_pending_signal_errors():
    """signal_tests.jl:834-917 / signal.jl:399-405: computing a non-pending signal is an ArgumentError.
    Through the engine: a final-round marginal that is pending computes; nothing else does."""
    E = ref.Engine(ref.P_TRACING)
    p = E.add_variable()
    f = E.add_factor(ref.F_OPAQUE)
    E.add_edge(p, f)
    E.finalize()
    E.update_marginals([p])  # nothing pending: no error, no execution
    assert E.counters() == (0, 0)


# ------------------------------------------------------------------ test/dependencies_tests.jl

def test_default_dependency_resolution_3var_2factor():
    """dependencies_tests.jl:39-99: exact dependency sets on v1 - f1 - v2 - f2 - v3."""
    E = ref.Engine()
    v1, v2, v3 = E.add_variable(), E.add_variable(), E.add_variable()
    f1, f2 = E.add_factor(), E.add_factor()
    for v, f in ((v1, f1), (v2, f1), (v2, f2), (v3, f2)):
        E.add_edge(v, f)
    E.finalize()
    m2v, m2f = E.message_to_variable, E.message_to_factor
    assert E.dependencies(E.marginal(v1)) == [m2v(v1, f1)]
    assert sorted(E.dependencies(E.marginal(v2))) == sorted([m2v(v2, f1), m2v(v2, f2)])
    assert E.dependencies(E.marginal(v3)) == [m2v(v3, f2)]
    assert E.dependencies(m2v(v2, f1)) == [m2f(v1, f1)]
    assert E.dependencies(m2v(v2, f2)) == [m2f(v3, f2)]
    assert E.dependencies(m2f(v2, f1)) == [m2v(v2, f2)]
    assert E.dependencies(m2f(v2, f2)) == [m2v(v2, f1)]


# ------------------------------------------------------------------ test/inference_engine_tests.jl

def test_warning_for_isolated_variable():
    """inference_engine_tests.jl:37-51."""
    E = ref.Engine()
    v = E.add_variable()
    E.finalize()
    assert E.warnings() == [v]


def test_signal_variants_prepared():
    """inference_engine_tests.jl:53-91."""
    E = ref.Engine()
    v1, v2, v3 = E.add_variable(), E.add_variable(), E.add_variable()
    f1, f2 = E.add_factor(), E.add_factor()
    for v, f in ((v1, f1), (v2, f2), (v3, f1), (v3, f2)):
        E.add_edge(v, f)
    E.finalize()
    for v, f in ((v1, f1), (v2, f2), (v3, f1), (v3, f2)):
        assert E.variant(E.message_to_variable(v, f))[:3] == (ref.VAR_MSG_TO_VARIABLE, v, f)
        assert E.variant(E.message_to_factor(v, f))[:3] == (ref.VAR_MSG_TO_FACTOR, v, f)
    for v in (v1, v2, v3):
        assert E.variant(E.marginal(v))[:2] == (ref.VAR_MARGINAL, v)


def _small_node_variable_node():
    E = ref.Engine()
    f1, f2 = E.add_factor(), E.add_factor()
    vc = E.add_variable()
    E.add_edge(vc, f1)
    E.add_edge(vc, f2)
    return E, f1, f2, vc


def test_empty_scan():
    """inference_engine_tests.jl:93-114."""
    E, f1, f2, vc = _small_node_variable_node()
    E.finalize()
    assert E.scan([vc]) == []


def test_scan_collects_pending_messages_in_order():
    """inference_engine_tests.jl:116-181: 1 / 1 / 2 steps, order [f1->vc, f2->vc]."""
    def make():
        E, f1, f2, vc = _small_node_variable_node()
        E.finalize(resolve_dependencies=False)
        vm, left, right = E.marginal(vc), E.signal(), E.signal()
        E.add_dependency(E.message_to_variable(vc, f1), left)
        E.add_dependency(E.message_to_variable(vc, f2), right)
        E.add_dependency(vm, E.message_to_variable(vc, f1))
        E.add_dependency(vm, E.message_to_variable(vc, f2))
        return E, f1, f2, vc, left, right

    E, f1, f2, vc, left, right = make()
    E.set_value(left, 1.0)
    assert E.scan([vc]) == [E.message_to_variable(vc, f1)]
    E, f1, f2, vc, left, right = make()
    E.set_value(right, 1.0)
    assert E.scan([vc]) == [E.message_to_variable(vc, f2)]
    E, f1, f2, vc, left, right = make()
    E.set_value(left, 1.0)
    E.set_value(right, 1.0)
    assert E.scan([vc]) == [E.message_to_variable(vc, f1), E.message_to_variable(vc, f2)]


def test_scan_resolves_dependencies_of_required_messages():
    """inference_engine_tests.jl:183-239."""
    E = ref.Engine()
    v1, v2, v3 = E.add_variable(), E.add_variable(), E.add_variable()
    f1, f2 = E.add_factor(), E.add_factor()
    for v, f in ((v1, f1), (v2, f1), (v2, f2), (v3, f2)):
        E.add_edge(v, f)
    E.finalize(resolve_dependencies=False)
    E.add_dependency(E.message_to_variable(v2, f1), E.message_to_factor(v1, f1))
    E.add_dependency(E.message_to_variable(v2, f2), E.message_to_factor(v3, f2))
    E.add_dependency(E.marginal(v2), E.message_to_variable(v2, f1))
    E.add_dependency(E.marginal(v2), E.message_to_variable(v2, f2))
    E.set_value(E.message_to_factor(v1, f1), 1.0)
    E.set_value(E.message_to_factor(v3, f2), 1.0)
    assert E.scan([v2]) == [E.message_to_variable(v2, f1), E.message_to_variable(v2, f2)]


@pytest.mark.parametrize("n,seed", [(100, 1234), (7, 1), (6, 2), (5, 3), (33, 4)])
def test_beta_bernoulli_exact_posterior(n, seed):
    """inference_engine_tests.jl:241-377: posterior Beta(1 + Σ, 1 + n − Σ), through the segment-tree wiring
    (dependencies.jl:90-173) for n > 5 and the all-pairs wiring for n ≤ 5.  The reference draws its dataset from
    StableRNG(1234), which cannot be regenerated here; the known answer is data-independent in form."""
    rng = np.random.default_rng(seed)
    data = rng.random(n) < 0.5
    E = ref.Engine(ref.P_BETA_BERNOULLI)
    p = E.add_variable()
    o, f = [], []
    for _ in range(n):
        oi, fi = E.add_variable(), E.add_factor(ref.F_BERNOULLI)
        o.append(oi); f.append(fi)
        E.add_edge(p, fi)
        E.add_edge(oi, fi)
    E.finalize()
    for i in range(n):
        E.set_value(E.message_to_factor(o[i], f[i]), bool(data[i]))
    E.update_marginals([p])
    tag, a, b = E.get_value(E.marginal(p))
    assert tag == ref.BETA
    assert a == pytest.approx(1.0 + data.sum()) and b == pytest.approx(1.0 + n - data.sum())
    if n > 5:  # segment tree: the marginal is the product of two ProductOfMessages halves (dependencies.jl:124-125)
        deps = E.dependencies(E.marginal(p))
        assert len(deps) == 2
        kinds = {E.variant(d)[0] for d in deps}
        assert kinds <= {ref.VAR_PRODUCT, ref.VAR_MSG_TO_VARIABLE}


def test_ssm_belief_propagation_n100():
    """inference_engine_tests.jl:379-488: the reference asserts means ≥ 0, non-decreasing, variances ≥ 0 for
    data 2i + randn; checked here, plus equality with the exact smoother the reference does not test."""
    n = 100
    rng = np.random.default_rng(1234)
    data = 2.0 * np.arange(1, n + 1) + rng.standard_normal(n)
    E = ref.Engine(ref.P_SSM_BP)
    x = [E.add_variable() for _ in range(n)]
    y = [E.add_variable() for _ in range(n)]
    lik = [E.add_factor(ref.F_GAUSS_ADD, 1.0) for _ in range(n)]
    tr = [E.add_factor(ref.F_GAUSS_ADD, 1.0) for _ in range(n - 1)]
    assert (x[0], y[0], lik[0], tr[0], tr[-1]) == (1, n + 1, 2 * n + 1, 3 * n + 1, 4 * n - 1)
    for i in range(n):
        E.add_edge(y[i], lik[i])
        E.add_edge(x[i], lik[i])
    for i in range(n - 1):
        E.add_edge(x[i], tr[i])
        E.add_edge(x[i + 1], tr[i])
    E.finalize()
    for i in range(n):
        E.set_value(E.message_to_factor(y[i], lik[i]), float(data[i]))
    E.update_marginals(x)
    tags, mean, var = E.get_marginals(x)
    assert np.all(tags == ref.NORMAL)
    assert np.all(mean >= 0) and np.all(np.diff(mean) >= 0) and np.all(var >= 0)
    assert E.counters() == (5 * n - 4, n)  # 5T−4 message computations + T marginals
    em, ev = exact.ssm_chain_posterior(data, 1.0, 1.0)
    np.testing.assert_allclose(mean, em, rtol=1e-12)
    np.testing.assert_allclose(var, ev, rtol=1e-12)


def test_ssm_execution_order_T3():
    """SURVEY.md §3.3 hand trace of update_marginals! on the T=3 chain (what `trace=true` records)."""
    n = 3
    E = ref.Engine(ref.P_SSM_BP, trace=True)
    x = [E.add_variable() for _ in range(n)]
    y = [E.add_variable() for _ in range(n)]
    lik = [E.add_factor(ref.F_GAUSS_ADD, 1.0) for _ in range(n)]
    tr = [E.add_factor(ref.F_GAUSS_ADD, 1.0) for _ in range(n - 1)]
    for i in range(n):
        E.add_edge(y[i], lik[i]); E.add_edge(x[i], lik[i])
    for i in range(n - 1):
        E.add_edge(x[i], tr[i]); E.add_edge(x[i + 1], tr[i])
    E.finalize()
    for i in range(n):
        E.set_value(E.message_to_factor(y[i], lik[i]), float(i))
    E.update_marginals(x)
    m2v, m2f, M = E.message_to_variable, E.message_to_factor, E.marginal
    expect = [
        (0, m2v(x[0], lik[0])), (0, m2v(x[1], lik[1])), (0, m2f(x[0], tr[0])), (0, m2v(x[1], tr[0])),
        (0, m2v(x[2], lik[2])), (0, m2f(x[1], tr[1])), (0, m2v(x[2], tr[1])),
        (1, m2f(x[2], tr[1])), (1, m2v(x[1], tr[1])), (1, m2f(x[1], tr[0])), (1, m2v(x[0], tr[0])),
        (2, M(x[0])), (2, M(x[1])), (2, M(x[2])),
    ]
    assert [(r, s) for r, _v, s, _b, _a in E.trace()] == expect
    assert E.trace_rounds() == 3


def test_tracing_known_answer():
    """inference_engine_tests.jl:1149-1261: rounds = 2, executions [MsgToVar(p,f1), MsgToVar(p,f2)] then
    [IndividualMarginal(p)], values 2, 4, 9, value_before = UndefValue()."""
    E = ref.Engine(ref.P_TRACING, trace=True)
    p, o1, o2 = E.add_variable(), E.add_variable(), E.add_variable()
    fp = E.add_factor(ref.F_OPAQUE)
    f1, f2 = E.add_factor(ref.F_DOUBLE), E.add_factor(ref.F_DOUBLE)
    for v, f in ((p, fp), (p, f1), (p, f2), (o1, f1), (o2, f2)):
        E.add_edge(v, f)
    E.finalize()
    E.set_value(E.message_to_factor(o1, f1), 1.0)
    E.set_value(E.message_to_factor(o2, f2), 2.0)
    E.set_value(E.message_to_variable(p, fp), 3.0)
    E.update_marginals([p])
    assert E.get_value(E.marginal(p))[:2] == (ref.REAL, 9.0)
    tr = E.trace()
    assert E.trace_rounds() == 2
    assert [(r, v, s) for r, v, s, _b, _a in tr] == [(0, p, E.message_to_variable(p, f1)), (0, p, E.message_to_variable(p, f2)),
                                                     (1, p, E.marginal(p))]
    assert [b[0] for *_x, b, _a in tr] == [ref.UNDEF] * 3
    assert [a[1] for *_x, a in tr] == [2.0, 4.0, 9.0]


# ------------------------------------------------------------------ flooding restatement vs scheduler restatement

@pytest.mark.parametrize("T", [2, 3, 17, 64])
def test_flooding_order_reaches_scheduler_fixed_point_on_chain(T):
    """bp_flood.c (device order) and cortex_ref.c (reference order) agree on a tree, and with the exact smoother."""
    import cortex.jl_amd as cx
    from tests.helpers import engine_oracle_from_model, flood_oracle_from_model

    model = cx.synth.ssm_chain(T, seed=5, random_variances=True)
    g = flood_oracle_from_model(model)
    for _ in range(T + 2):
        g.sweep(1)
    E = engine_oracle_from_model(model)
    E.update_marginals(model.x_ids)
    _, em, ev = E.get_marginals(model.x_ids)
    gm, gv = g.marginals()
    xs = np.searchsorted(g.var_ids, model.x_ids)
    np.testing.assert_allclose(gm[xs], em, rtol=1e-12)
    np.testing.assert_allclose(gv[xs], ev, rtol=1e-12)
    xm, xv = exact.ssm_chain_posterior(model.data_y, model.meta["r"], model.meta["q"])
    np.testing.assert_allclose(em, xm, rtol=1e-11)
    np.testing.assert_allclose(ev, xv, rtol=1e-11)


def test_reference_order_and_flooding_order_share_the_loopy_fixed_point():
    """On a loopy grid the reference's sequential schedule (priors re-set before every update_marginals!, as a user
    must to make them fresh, signal.jl:668-730) and the flooding schedule converge to the same messages; the
    converged means are the exact posterior means."""
    import cortex.jl_amd as cx
    from tests.helpers import engine_oracle_from_model, flood_oracle_from_model

    model = cx.synth.gaussian_grid(8, 9, seed=5)
    E = engine_oracle_from_model(model)
    g = flood_oracle_from_model(model, 1e6)
    pe = g.partner >= 0
    E.set_messages_to_variable(g.edge_var[pe], g.edge_fac[pe], np.zeros(pe.sum()), np.full(pe.sum(), 1e6))
    for it in range(250):
        if it:
            E.set_messages_to_variable(model.prior_var, model.prior_fac, model.prior_mean, model.prior_variance)
        c0 = E.counters()[0]
        E.update_marginals(model.x_ids)
        assert E.counters()[0] - c0 == 2 * pe.sum()  # one sweep = every listened message once
    g.sweep(500)
    _, a, b = E.get_messages(g.edge_var, g.edge_fac, True)
    np.testing.assert_allclose(a, g.f2v_m, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(b, g.f2v_v, rtol=1e-10)
    _, em, _ev = E.get_marginals(model.x_ids)
    me = exact.grid_posterior_mean(8, 9, model.meta["y"], model.meta["r"], model.meta["qh"], model.meta["qv"])
    np.testing.assert_allclose(em, me, rtol=1e-9, atol=1e-11)
