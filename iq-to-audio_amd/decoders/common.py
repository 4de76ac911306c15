"""Shared demodulator pieces running on the HIP library."""
from __future__ import annotations

import math
from ctypes import c_double, c_int64, c_void_p

import numpy as np

from .. import _dev as D
from .. import _native as N


def scan_workspace(n: int):
    return D.empty(max(1, int(N.lib().iqa_scan_workspace_bytes(int(n)))), "uint8")


def rms_dbfs_of(audio_dev) -> float:
    """20*log10(sqrt(mean(a^2) + 1e-18) + 1e-12) with a float64 device reduction
    (reference decoders/nfm.py:88-89)."""
    n = int(audio_dev.numel())
    if n == 0:
        return float("nan")
    sumsq = D.zeros(8, "float64")  # IQA_SUMSQ_SLOTS sub-slots
    seg = D.zeros(1, "int64")
    N.call("iqa_writer_clip", N.ptr(audio_dev), c_int64(n), c_void_p(0), N.ptr(seg), c_int64(1), N.ptr(sumsq),
           c_void_p(0), N.stream_ptr())
    rms = math.sqrt(float(sumsq.sum().item()) / n + 1e-18)
    return 20.0 * math.log10(rms + 1e-12)


class DCBlocker:
    """One-pole DC blocker y[n] = x[n] - x[n-1] + r*y[n-1] (reference decoders/common.py:6-30),
    evaluated as a parallel affine scan on the GPU."""

    def __init__(self, radius: float = 0.995):
        if not 0.0 < radius < 1.0:
            raise ValueError("radius must be between 0 and 1")
        self.radius = radius
        self._state = None  # device double[2] = {x_prev, y_prev}

    def _ensure(self):
        if self._state is None:
            self._state = D.zeros(2, "float64")

    @property
    def _x_prev(self) -> float:
        return 0.0 if self._state is None else float(self._state[0].item())

    @property
    def _y_prev(self) -> float:
        return 0.0 if self._state is None else float(self._state[1].item())

    def process(self, samples):
        if (samples.numel() if D.is_tensor(samples) else np.asarray(samples).size) == 0:
            return samples
        self._ensure()
        x = D.to_device(samples, "float32")
        y = D.empty(x.numel(), "float32")
        work = scan_workspace(x.numel())
        N.call("iqa_dc_block", N.ptr(x), c_int64(x.numel()), c_double(self.radius), N.ptr(self._state), N.ptr(y),
               N.ptr(work), N.stream_ptr())
        return D.like_input(y, samples)
