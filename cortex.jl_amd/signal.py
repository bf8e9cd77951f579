"""Host-side mirror of the reference's reactive Signal runtime (src/signal.jl), which the host keeps.

Same names (minus Julia's `!`), same argument meaning, same error behaviour: `set_value!` → set_value,
`add_dependency!` → add_dependency, `compute!` → compute, `process_dependencies!` → process_dependencies.
Readiness is tracked exactly as in src/signal.jl:17-45,507-730: 4 bits per dependency
(Intermediate 0x1, Weak 0x2, Computed 0x4, Fresh 0x8), 16 dependencies per 64-bit chunk, and the pending rule
"every dependency: Computed & (Weak | Fresh)" evaluated chunk-parallel with the same masks.
"""
from __future__ import annotations

from typing import Any, Callable, List


class UndefValue:
    """signal.jl:7 — singleton: the signal has not been computed."""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
        return cls._inst

    def __repr__(self):
        return "UndefValue()"


class UndefVariant:
    """signal.jl:15"""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = super().__new__(cls)
        return cls._inst

    def __repr__(self):
        return "UndefVariant()"


_M64 = 0xFFFFFFFFFFFFFFFF
IS_INTERMEDIATE, IS_WEAK, IS_COMPUTED, IS_FRESH = 0x1, 0x2, 0x4, 0x8          # signal.jl:507-510
ALL_WEAK, ALL_COMPUTED, ALL_FRESH = 0x2222222222222222, 0x4444444444444444, 0x8888888888888888
ALL_PASS = 0x1111111111111111                                                 # signal.jl:519


class SignalDependenciesProps:
    """signal.jl:36-45: bit-packed dependency properties; always owns at least one chunk."""
    __slots__ = ("length", "chunks")

    def __init__(self):
        self.length = 0
        self.chunks = [0]

    def add(self) -> int:
        """signal.jl:529-544; returns the (0-based) index of the new nibble."""
        self.length += 1
        need = (4 * self.length - 1) // 64 + 1
        if len(self.chunks) < need:
            self.chunks.append(0)
        return self.length - 1

    def set(self, i: int, mask: int):
        self.chunks[i >> 4] |= (mask << ((i & 15) << 2)) & _M64

    def get(self, i: int, mask: int) -> bool:
        return (self.chunks[i >> 4] & (mask << ((i & 15) << 2))) != 0

    def unset_all_fresh(self):
        """signal.jl:653-655"""
        for c in range(len(self.chunks)):
            self.chunks[c] &= ~ALL_FRESH & _M64

    def is_meeting_pending_criteria(self) -> bool:
        """signal.jl:668-730"""
        n = self.length
        if n == 0:
            return False
        for c in range(len(self.chunks) - 1):
            ch = self.chunks[c]
            W, Cb, F = (ch & ALL_WEAK) >> 1, (ch & ALL_COMPUTED) >> 2, (ch & ALL_FRESH) >> 3
            if (Cb & (W | F)) != ALL_PASS:
                return False
        last, off = (n - 1) >> 4, ((n - 1) & 15) << 2
        fill = (_M64 << (off + 4)) & _M64          # Julia's UInt64 << 64 is 0, like the masked Python shift
        ch = self.chunks[last] | fill
        W, Cb, F = (ch & ALL_WEAK) >> 1, (ch & ALL_COMPUTED) >> 2, (ch & ALL_FRESH) >> 3
        return (Cb & (W | F)) == ALL_PASS


class Signal:
    """signal.jl:82-115.  value / variant / props / dependencies_props / dependencies / listenmask / listeners."""
    __slots__ = ("value", "variant", "is_potentially_pending", "_is_pending", "dependencies_props", "dependencies",
                 "listenmask", "listeners")

    def __init__(self, value: Any = UndefValue(), variant: Any = UndefVariant()):
        self.value = value
        self.variant = variant
        self.is_potentially_pending = False
        self._is_pending = False
        self.dependencies_props = SignalDependenciesProps()
        self.dependencies: List[Signal] = []
        self.listenmask: List[bool] = []
        self.listeners: List[Signal] = []

    def __repr__(self):  # signal.jl:360-370
        val = repr(self.value) if is_computed(self) else "#undef"
        s = f"Signal(value={val}, pending={'true' if is_pending(self) else 'false'}"
        if not isinstance(self.variant, UndefVariant):
            s += f", variant={self.variant!r}"
        return s + ")"


def is_pending(s: Signal) -> bool:
    """signal.jl:141-154"""
    if s._is_pending:
        return True
    if s.is_potentially_pending:
        p = s.dependencies_props.is_meeting_pending_criteria()
        s.is_potentially_pending = False
        s._is_pending = p
        return p
    return False


def is_computed(s: Signal) -> bool:
    """signal.jl:162-164"""
    return not isinstance(s.value, UndefValue)


def get_value(s: Signal):
    return s.value


def get_variant(s: Signal):
    return s.variant


def set_variant(s: Signal, variant):
    s.variant = variant


def isa_variant(s: Signal, T) -> bool:
    return isinstance(s.variant, T)


def get_dependencies(s: Signal):
    return s.dependencies


def get_listeners(s: Signal):
    return s.listeners


def _notify_listener(listener: Signal, signal: Signal, update_potentially_pending: bool):
    """signal.jl:339-356"""
    if update_potentially_pending:
        listener.is_potentially_pending, listener._is_pending = True, False
    for i, d in enumerate(listener.dependencies):
        if d is signal:  # duplicates never receive a notification (:344)
            listener.dependencies_props.set(i, IS_FRESH)
            listener.dependencies_props.set(i, IS_COMPUTED)
            break


def set_value(signal: Signal, value):
    """signal.jl:232-253"""
    signal.value = value
    signal.dependencies_props.unset_all_fresh()
    signal.is_potentially_pending, signal._is_pending = False, False
    for is_listening, listener in zip(signal.listenmask, signal.listeners):
        _notify_listener(listener, signal, is_listening)


def add_dependency(signal: Signal, dependency: Signal, *, weak: bool = False, listen: bool = True,
                   check_computed: bool = True, intermediate: bool = False):
    """signal.jl:286-337"""
    if signal is dependency:
        return
    props = signal.dependencies_props
    idx = props.add()
    if weak:
        props.set(idx, IS_WEAK)
    if intermediate:
        props.set(idx, IS_INTERMEDIATE)
    signal.dependencies.append(dependency)
    dependency.listenmask.append(bool(listen))
    dependency.listeners.append(signal)
    if check_computed and is_computed(dependency):
        props.set(idx, IS_COMPUTED)
        if not is_computed(signal):
            props.set(idx, IS_FRESH)
        signal.is_potentially_pending, signal._is_pending = True, False
    elif check_computed:
        signal.is_potentially_pending, signal._is_pending = False, False


def compute(strategy: Callable, signal: Signal, *, force: bool = False, skip_if_no_listeners: bool = False):
    """signal.jl:392-410: `compute!(strategy, signal)`; raises ValueError (ArgumentError) on a non-pending signal."""
    if skip_if_no_listeners and not signal.listeners:
        return
    if not force and not is_pending(signal):
        raise ValueError("Signal is not pending. Cannot compute a non-pending signal. Use `force=true` to force "
                         f"computation. Signal: {signal!r}")
    new_value = strategy(signal, signal.dependencies)
    set_value(signal, new_value)


def process_dependencies(f: Callable[[Signal], bool], signal: Signal, *, retry: bool = False) -> bool:
    """signal.jl:466-490"""
    any_ = False
    props = signal.dependencies_props
    for i, dep in enumerate(signal.dependencies):
        processed = bool(f(dep))
        if not processed and props.get(i, IS_INTERMEDIATE):
            sub = process_dependencies(f, dep, retry=retry)
            if sub and retry:
                processed = bool(f(dep))
            any_ = any_ or sub
        any_ = any_ or processed
    return any_
