"""Narrow-band FM: phase-difference discriminator + single-pole de-emphasis
(reference decoders/nfm.py), on the HIP library."""
from __future__ import annotations

import math
from ctypes import c_double, c_int64

import numpy as np

from .. import _dev as D
from .. import _native as N
from .base import Decoder, DecoderStats
from .common import rms_dbfs_of, scan_workspace


def _size(x) -> int:
    return int(x.numel()) if D.is_tensor(x) else int(np.asarray(x).size)


class QuadratureDemod:
    """angle(z[n] * conj(z[n-1])) in radians/sample (reference decoders/nfm.py:11-24)."""

    def __init__(self):
        self._prev = None  # device complex64[1], starts at 1+0j

    @property
    def prev(self):
        return np.complex64(1 + 0j) if self._prev is None else np.complex64(self._prev.cpu().numpy()[0])

    def process(self, samples):
        if _size(samples) == 0:
            return np.empty(0, dtype=np.float32) if not D.is_tensor(samples) else D.empty(0, "float32")
        if self._prev is None:
            self._prev = D.from_numpy(np.array([1 + 0j], dtype=np.complex64))
        z = D.to_device(samples, "complex64")
        out = D.empty(z.numel(), "float32")
        N.call("iqa_quadrature", N.ptr(z), c_int64(z.numel()), N.ptr(self._prev), N.ptr(out), N.stream_ptr())
        return D.like_input(out, samples)


class DeemphasisFilter:
    """y[n] = (1-a) x[n] + a y[n-1], a = exp(-1/(fs*tau)) (reference decoders/nfm.py:27-62)."""

    def __init__(self, tau_us: float, sample_rate: float | None = None):
        self.tau_us = tau_us
        self.alpha = 0.0
        self.beta = 0.0
        self._y_last = None  # device double[1]
        if sample_rate is not None:
            self.configure(sample_rate)

    def configure(self, sample_rate: float) -> None:
        tau_sec = max(self.tau_us * 1e-6, 1e-6)
        self.alpha = math.exp(-1.0 / (sample_rate * tau_sec))
        self.beta = 1.0 - self.alpha
        self._y_last = None

    @property
    def state(self) -> float:
        """The reference's carried value: scipy lfilter's zf = alpha * y[last]."""
        return 0.0 if self._y_last is None else self.alpha * float(self._y_last.item())

    def process(self, samples):
        if _size(samples) == 0:
            return samples
        if self._y_last is None:
            self._y_last = D.zeros(1, "float64")
        x = D.to_device(samples, "float32")
        y = D.empty(x.numel(), "float32")
        work = scan_workspace(x.numel())
        N.call("iqa_deemphasis", N.ptr(x), c_int64(x.numel()), c_double(self.alpha), N.ptr(self._y_last), N.ptr(y),
               N.ptr(work), N.stream_ptr())
        return D.like_input(y, samples)


class NarrowbandFMDecoder(Decoder):
    """Discriminator then de-emphasis; AGC flag ignored (reference decoders/nfm.py:65-108)."""

    name = "narrowband_fm"

    def __init__(self, deemph_us: float):
        self._deemph_us = deemph_us
        self._demod = QuadratureDemod()
        self._deemph = DeemphasisFilter(deemph_us)
        self._last_stats = None
        self._intermediates = {}
        self._sample_rate = 0.0

    def setup(self, sample_rate: float) -> None:
        self._deemph.configure(sample_rate)
        self._sample_rate = sample_rate

    def process(self, samples):
        if self._sample_rate == 0.0:
            raise RuntimeError("Decoder.setup(sample_rate) must be called before processing data.")
        z = D.to_device(samples, "complex64")
        demod = self._demod.process(z)
        deemph = self._deemph.process(demod)
        stats = DecoderStats(rms_dbfs=rms_dbfs_of(deemph))
        self._last_stats = stats
        if z.numel():
            self._intermediates = {"demod": (demod, self._sample_rate), "deemph": (deemph, self._sample_rate),
                                   "audio": (deemph, self._sample_rate)}
        return D.like_input(deemph, samples), stats

    def finalize(self) -> None:
        return

    @property
    def last_stats(self):
        return self._last_stats

    def intermediates(self) -> dict:
        """Stage buffers; copied to the host lazily, here (D2H only when asked)."""
        return {k: (v.cpu().numpy().copy(), r) for k, (v, r) in self._intermediates.items()}
