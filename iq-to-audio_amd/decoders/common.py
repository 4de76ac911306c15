"""The DC blocker shared by the AM and SSB decoders, on the HIP library."""
from __future__ import annotations

from ctypes import c_double, c_int64

from .. import _dev as D
from .. import _native as N
from .base import n_elements, scan_workspace


class DCBlocker:
    """y[n] = x[n] - x[n-1] + r y[n-1] across calls (reference decoders/common.py:6-30), as a parallel affine scan in
    float64; the pair (x[last], y[last]) lives in a device double[2]."""

    def __init__(self, radius: float = 0.995):
        if not 0.0 < radius < 1.0:
            raise ValueError("radius must be between 0 and 1")
        self.radius = radius
        self._pair = None

    def carried(self) -> tuple[float, float]:
        """(x_prev, y_prev) as the next call will see them."""
        if self._pair is None:
            return 0.0, 0.0
        x_prev, y_prev = self._pair.cpu().numpy()
        return float(x_prev), float(y_prev)

    def process(self, samples):
        if n_elements(samples) == 0:
            return samples
        if self._pair is None:
            self._pair = D.zeros(2, "float64")
        x = D.to_device(samples, "float32")
        y = D.empty(x.numel(), "float32")
        N.call("iqa_dc_block", N.ptr(x), c_int64(x.numel()), c_double(self.radius), N.ptr(self._pair), N.ptr(y),
               N.ptr(scan_workspace(x.numel())), N.stream_ptr())
        return D.like_input(y, samples)
