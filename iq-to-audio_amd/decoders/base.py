"""Decoder plug-in interface (same contract as the reference's ``decoders/base.py:9-37``)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass


@dataclass
class DecoderStats:
    """Runtime statistics from a decoder stage (reference decoders/base.py:9-13)."""

    rms_dbfs: float


class Decoder(ABC):
    """Abstract demodulator: ``setup(rate)``, ``process(samples) -> (audio, stats)``,
    ``finalize()``, ``intermediates()``.  ``samples`` may be a NumPy complex64 array (NumPy
    comes back, as in the reference) or a device tensor (device tensors come back)."""

    name: str = "decoder"

    @abstractmethod
    def setup(self, sample_rate: float) -> None:
        """Prepare decoder state for the given input sample rate."""

    @abstractmethod
    def finalize(self) -> None:
        """Allow decoder to flush any pending state."""

    @abstractmethod
    def process(self, samples):
        """Consume baseband samples and return audio plus optional stats."""

    def intermediates(self) -> dict:
        """Diagnostic intermediate buffers keyed by stage name."""
        return {}
