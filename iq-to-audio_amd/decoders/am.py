"""AM: envelope detector + DC blocker (reference decoders/am.py), on the HIP library."""
from __future__ import annotations

from ctypes import c_int64

from .. import _dev as D
from .. import _native as N
from .base import Decoder, DecoderStats
from .common import DCBlocker, rms_dbfs_of


class AMDecoder(Decoder):
    """|z| then y[n] = x[n]-x[n-1]+r*y[n-1]; no AGC, no normalisation (reference decoders/am.py:11-50)."""

    name = "am"

    def __init__(self, dc_radius: float = 0.995):
        self._dc_blocker = DCBlocker(radius=dc_radius)
        self._last_stats = None
        self._intermediates = {}
        self._sample_rate = 0.0

    def setup(self, sample_rate: float) -> None:
        self._sample_rate = sample_rate

    def process(self, samples):
        if self._sample_rate == 0.0:
            raise RuntimeError("Decoder.setup(sample_rate) must be called before processing data.")
        z = D.to_device(samples, "complex64")
        envelope = D.empty(z.numel(), "float32")
        N.call("iqa_envelope", N.ptr(z), c_int64(z.numel()), N.ptr(envelope), N.stream_ptr())
        ac_coupled = self._dc_blocker.process(envelope)
        stats = DecoderStats(rms_dbfs=rms_dbfs_of(ac_coupled))
        self._last_stats = stats
        if z.numel():
            self._intermediates = {"envelope": (envelope, self._sample_rate), "dc_block": (ac_coupled, self._sample_rate),
                                   "audio": (ac_coupled, self._sample_rate)}
        return D.like_input(ac_coupled, samples), stats

    def finalize(self) -> None:
        return

    def intermediates(self) -> dict:
        return {k: (v.cpu().numpy().copy(), r) for k, (v, r) in self._intermediates.items()}

    @property
    def last_stats(self):
        return self._last_stats
