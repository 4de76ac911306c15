"""Progress / cancel boundary types (the six-method sink contract of the reference's
``progress.py:13-78,159-242``).  Pure host bookkeeping; the pipeline calls it between blocks."""
from __future__ import annotations

from collections.abc import Callable, Iterable
from dataclasses import dataclass


@dataclass
class PhaseState:
    key: str
    label: str
    total: float
    unit: str = "samples"
    completed: float = 0.0

    def remaining(self) -> float:
        return max(self.total - self.completed, 0.0)


class ProgressSink:
    """Interface for receiving progress events."""

    def start(self, phases: Iterable[PhaseState], *, overall_total: float) -> None:
        raise NotImplementedError

    def advance(self, phase: PhaseState, delta: float, *, overall_completed: float, overall_total: float) -> None:
        raise NotImplementedError

    def status(self, message: str) -> None:
        raise NotImplementedError

    def close(self) -> None:
        raise NotImplementedError

    def set_cancel_callback(self, callback: Callable[[], None]) -> None:
        return

    def cancel(self) -> None:
        raise NotImplementedError


class NullProgressSink(ProgressSink):
    def start(self, phases, *, overall_total):
        return

    def advance(self, phase, delta, *, overall_completed, overall_total):
        return

    def status(self, message):
        return

    def close(self):
        return

    def cancel(self):
        return


class ProgressTracker:
    """Fans pipeline events out to a sink and keeps per-phase totals."""

    def __init__(self, sink: ProgressSink | None):
        self.sink = sink or NullProgressSink()
        self.phases: dict[str, PhaseState] = {}
        self.cancelled = False
        self._total = 0.0
        self._done = 0.0

    def start(self, phases: Iterable[PhaseState]) -> None:
        plist = list(phases)
        self.phases = {p.key: p for p in plist}
        self._total = float(sum(max(p.total, 0.0) for p in plist))
        self._done = 0.0
        self.sink.start(plist, overall_total=self._total)

    def advance(self, key: str, delta: float) -> None:
        phase = self.phases.get(key)
        if phase is None or delta <= 0:
            return
        phase.completed += delta
        self._done += delta
        self.sink.advance(phase, delta, overall_completed=self._done, overall_total=self._total)

    def status(self, message: str) -> None:
        self.sink.status(message)

    def cancel(self) -> None:
        self.cancelled = True
        try:
            self.sink.cancel()
        except NotImplementedError:
            pass

    def close(self) -> None:
        self.sink.close()
