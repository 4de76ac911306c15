"""AM: envelope, then the DC blocker (reference decoders/am.py:11-50) -- no AGC, no normalisation."""
from __future__ import annotations

from ctypes import c_int64

from .. import _dev as D
from .. import _native as N
from .base import GpuDecoder
from .common import DCBlocker


class AMDecoder(GpuDecoder):
    name = "am"

    def __init__(self, dc_radius: float = 0.995):
        super().__init__()
        self.dc = DCBlocker(radius=dc_radius)

    def stages(self, z) -> list:
        mag = D.empty(z.numel(), "float32")
        N.call("iqa_envelope", N.ptr(z), c_int64(z.numel()), N.ptr(mag), N.stream_ptr())
        return [("envelope", mag), ("dc_block", self.dc.process(mag))]

    def fused_params(self) -> N.DemodParams:
        return N.DemodParams(mode=N.DEMOD_MODE["am"], agc_enabled=0, deemph_alpha=0.0, dc_radius=self.dc.radius,
                             agc_target=0.0, agc_decay=0.0)
