"""Progress / cancel boundary of the pipeline.

What a caller implements is the reference's sink contract (``progress.py:13-78``): a ``PhaseState`` record per phase and
a sink with ``start / advance / status / close / set_cancel_callback / cancel``.  The reference's own sinks (tqdm bars,
the Qt bridge) are UI and out of scope; here are the contract, a sink that ignores everything, and the small fan-out
object the pipeline drives between device blocks -- pure host bookkeeping.
"""
from __future__ import annotations

from collections.abc import Callable, Iterable
from dataclasses import dataclass


@dataclass
class PhaseState:
    """One phase of a run as the sink sees it (field set of the reference's record)."""

    key: str
    label: str
    total: float
    unit: str = "samples"
    completed: float = 0.0

    def remaining(self) -> float:
        return self.total - self.completed if self.completed < self.total else 0.0


class ProgressSink:
    """Receiver of progress events.  A sink overrides what it needs; the three reporting methods and ``close`` /
    ``cancel`` have no default behaviour on purpose (a sink that forgets one fails loudly, as in the reference)."""

    def start(self, phases: Iterable[PhaseState], *, overall_total: float) -> None:
        raise NotImplementedError

    def advance(self, phase: PhaseState, delta: float, *, overall_completed: float, overall_total: float) -> None:
        raise NotImplementedError

    def status(self, message: str) -> None:
        raise NotImplementedError

    def close(self) -> None:
        raise NotImplementedError

    def cancel(self) -> None:
        raise NotImplementedError

    def set_cancel_callback(self, callback: Callable[[], None]) -> None:
        """Optional: a sink with a cancel control keeps ``callback`` and calls it when the user asks to stop."""


class NullProgressSink(ProgressSink):
    """Swallows every event (what ``run(progress_sink=None)`` uses)."""

    def start(self, phases, *, overall_total):
        pass

    def advance(self, phase, delta, *, overall_completed, overall_total):
        pass

    def status(self, message):
        pass

    def close(self):
        pass

    def cancel(self):
        pass


class ProgressTracker:
    """The pipeline's side of the contract: phases by key, running totals, and the cancelled flag a sink may raise."""

    def __init__(self, sink: ProgressSink | None):
        self.sink = sink if sink is not None else NullProgressSink()
        self.phases: dict[str, PhaseState] = {}
        self.cancelled = False
        self._done = self._total = 0.0

    def start(self, phases: Iterable[PhaseState]) -> None:
        listed = list(phases)
        self.phases = {ph.key: ph for ph in listed}
        self._done, self._total = 0.0, float(sum(ph.total for ph in listed if ph.total > 0))
        self.sink.start(listed, overall_total=self._total)

    def advance(self, key: str, delta: float) -> None:
        """``delta`` more units of phase ``key`` are done (unknown phases and non-positive deltas are ignored)."""
        phase = self.phases.get(key)
        if phase is not None and delta > 0:
            phase.completed += delta
            self._done += delta
            self.sink.advance(phase, delta, overall_completed=self._done, overall_total=self._total)

    def status(self, message: str) -> None:
        self.sink.status(message)

    def cancel(self) -> None:
        self.cancelled = True
        try:
            self.sink.cancel()
        except NotImplementedError:  # a sink without a cancel control
            pass

    def close(self) -> None:
        self.sink.close()
