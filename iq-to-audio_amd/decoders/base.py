"""Decoder plug-in contract and the shared body of the HIP decoders.

The public contract -- ``Decoder`` with ``name``, ``setup``, ``process``, ``finalize``, ``intermediates`` and
``DecoderStats.rms_dbfs`` -- is the reference's (decoders/base.py:9-37); everything a demodulator here has in common
lives once in :class:`GpuDecoder`: the rate check, the stage-by-stage evaluation on the device, the level statistic and
the lazily copied stage buffers.  A concrete decoder only lists its stages and says how the fused block path
(``iqa_demodulate``) should be parameterised for it.
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from ctypes import c_int64, c_void_p
from dataclasses import dataclass

import numpy as np

from .. import _dev as D
from .. import _native as N


@dataclass
class DecoderStats:
    """Level of one ``process`` call's output (reference decoders/base.py:9-13)."""

    rms_dbfs: float


class Decoder(ABC):
    """What the pipeline asks of a demodulator.  ``samples`` is a NumPy complex64 array (NumPy comes back, as in the
    reference) or a device tensor (a device tensor comes back, no host round trip)."""

    name: str = "decoder"

    @abstractmethod
    def setup(self, sample_rate: float) -> None:
        """Fix the input rate; must precede ``process``."""

    @abstractmethod
    def process(self, samples):
        """Baseband block in, ``(audio, DecoderStats)`` out."""

    @abstractmethod
    def finalize(self) -> None:
        """End of stream."""

    def intermediates(self) -> dict:
        """``{stage name: (array, rate)}`` of the most recent non-empty block."""
        return {}


def scan_workspace(n: int):
    """Device scratch of the three-pass scans for ``n`` elements."""
    return D.empty(max(1, int(N.lib().iqa_scan_workspace_bytes(int(n)))), "uint8")


def level_dbfs(audio_dev) -> float:
    """20 log10(sqrt(mean(a^2) + 1e-18) + 1e-12) of a device float32 stream, the mean from a float64 device reduction
    (the statistic every reference decoder reports: decoders/nfm.py:88-89, am.py:31-32, ssb.py:46-47)."""
    n = int(audio_dev.numel())
    if n == 0:
        return float("nan")
    sums = D.zeros(8, "float64")  # IQA_SUMSQ_SLOTS sub-slots of the single segment
    origin = D.zeros(1, "int64")
    N.call("iqa_writer_clip", N.ptr(audio_dev), c_int64(n), c_void_p(0), N.ptr(origin), c_int64(1), N.ptr(sums), c_void_p(0),
           N.stream_ptr())
    return 20.0 * math.log10(math.sqrt(float(sums.sum().item()) / n + 1e-18) + 1e-12)


class GpuDecoder(Decoder):
    """A demodulator as an ordered list of device stages.

    Subclasses implement ``stages(z) -> [(name, tensor), ...]`` (the last tensor is the audio), ``on_rate(rate)`` when
    a stage depends on the sample rate, and ``fused_params()`` -- the ``iqa_demod_params`` of the fused block path that
    computes the same chain for many reference chunks at once (``processing.ChannelDemod``).
    """

    def __init__(self) -> None:
        self.rate = 0.0
        self.stats: DecoderStats | None = None
        self._held: list = []  # [(stage name, device tensor)] of the last non-empty block

    # -- contract --------------------------------------------------------------------------------
    def setup(self, sample_rate: float) -> None:
        self.rate = float(sample_rate)
        self.on_rate(self.rate)

    def process(self, samples):
        if self.rate == 0.0:
            raise RuntimeError("Decoder.setup(sample_rate) must be called before processing data.")
        z = D.to_device(samples, "complex64")
        chain = self.stages(z)
        audio = chain[-1][1]
        self.stats = DecoderStats(rms_dbfs=level_dbfs(audio))
        if z.numel():
            self._held = chain + [("audio", audio)]
        return D.like_input(audio, samples), self.stats

    def finalize(self) -> None:
        return None

    def intermediates(self) -> dict:
        """Copied to the host here, on demand (nothing leaves the device during ``process``)."""
        return {name: (buf.cpu().numpy().copy(), self.rate) for name, buf in self._held}

    @property
    def last_stats(self) -> DecoderStats | None:
        return self.stats

    # -- for subclasses --------------------------------------------------------------------------
    def on_rate(self, rate: float) -> None:
        return None

    def stages(self, z) -> list:
        raise NotImplementedError

    def fused_params(self) -> "N.DemodParams":
        raise NotImplementedError


def n_elements(x) -> int:
    return int(x.numel()) if D.is_tensor(x) else int(np.asarray(x).size)
