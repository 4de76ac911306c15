"""Narrow-band FM: phase-difference discriminator, then a one-pole de-emphasis (reference decoders/nfm.py)."""
from __future__ import annotations

import math
from ctypes import c_double, c_int64

import numpy as np

from .. import _dev as D
from .. import _native as N
from .base import GpuDecoder, n_elements, scan_workspace


class QuadratureDemod:
    """angle(z[n] conj(z[n-1])) in radians per sample, z[-1] = 1 + 0j at the start of a stream and carried across calls
    (reference decoders/nfm.py:11-24)."""

    def __init__(self):
        self._last = None  # device complex64[1]

    @property
    def prev(self):
        return np.complex64(1 + 0j) if self._last is None else np.complex64(self._last.cpu().numpy()[0])

    def process(self, samples):
        if n_elements(samples) == 0:
            return D.empty(0, "float32") if D.is_tensor(samples) else np.empty(0, dtype=np.float32)
        if self._last is None:
            self._last = D.from_numpy(np.array([1 + 0j], dtype=np.complex64))
        z = D.to_device(samples, "complex64")
        out = D.empty(z.numel(), "float32")
        N.call("iqa_quadrature", N.ptr(z), c_int64(z.numel()), N.ptr(self._last), N.ptr(out), N.stream_ptr())
        return D.like_input(out, samples)


class DeemphasisFilter:
    """y[n] = (1 - a) x[n] + a y[n-1], a = exp(-1 / (fs tau)), tau = max(tau_us, 1) microseconds
    (reference decoders/nfm.py:27-62), as a float64 scan; y[last] lives in a device double."""

    def __init__(self, tau_us: float, sample_rate: float | None = None):
        self.tau_us = tau_us
        self.alpha = self.beta = 0.0
        self._y = None
        if sample_rate is not None:
            self.configure(sample_rate)

    def configure(self, sample_rate: float) -> None:
        self.alpha = math.exp(-1.0 / (sample_rate * max(self.tau_us * 1e-6, 1e-6)))
        self.beta = 1.0 - self.alpha
        self._y = None

    @property
    def state(self) -> float:
        """What the reference carries between calls: scipy ``lfilter``'s final condition, alpha * y[last]."""
        return 0.0 if self._y is None else self.alpha * float(self._y.item())

    def process(self, samples):
        if n_elements(samples) == 0:
            return samples
        if self._y is None:
            self._y = D.zeros(1, "float64")
        x = D.to_device(samples, "float32")
        y = D.empty(x.numel(), "float32")
        N.call("iqa_deemphasis", N.ptr(x), c_int64(x.numel()), c_double(self.alpha), N.ptr(self._y), N.ptr(y),
               N.ptr(scan_workspace(x.numel())), N.stream_ptr())
        return D.like_input(y, samples)


class NarrowbandFMDecoder(GpuDecoder):
    """Discriminator output in radians per sample (no deviation scaling), de-emphasised; the AGC switch does not apply
    to this mode (reference decoders/nfm.py:65-108, decoders/__init__.py:16-17)."""

    name = "narrowband_fm"

    def __init__(self, deemph_us: float):
        super().__init__()
        self.discriminator = QuadratureDemod()
        self.deemphasis = DeemphasisFilter(deemph_us)

    def on_rate(self, rate: float) -> None:
        self.deemphasis.configure(rate)

    def stages(self, z) -> list:
        phase = self.discriminator.process(z)
        return [("demod", phase), ("deemph", self.deemphasis.process(phase))]

    def fused_params(self) -> N.DemodParams:
        return N.DemodParams(mode=N.DEMOD_MODE["nfm"], agc_enabled=0, deemph_alpha=self.deemphasis.alpha, dc_radius=0.995,
                             agc_target=0.0, agc_decay=0.0)
