"""SSB: real part, DC blocker, optional per-call AGC (reference decoders/ssb.py:11-84).

Two properties of the reference are kept on purpose: USB and LSB give the same audio (the real part of z and of its
conjugate are equal, ssb.py:42-44), and the AGC gain starts again at 1.0 on every ``process`` call (:72) -- the
pipeline's fused path expresses those restarts as segment starts of one scan.
"""
from __future__ import annotations

from ctypes import c_double, c_int64, c_void_p

from .. import _dev as D
from .. import _native as N
from .base import GpuDecoder, scan_workspace
from .common import DCBlocker


class SSBDecoder(GpuDecoder):
    name = "ssb"

    def __init__(self, sideband: str, agc_enabled: bool, dc_radius: float = 0.995, agc_target_dbfs: float = -12.0,
                 agc_decay: float = 0.001):
        super().__init__()
        side = sideband.lower()
        if side not in ("usb", "lsb"):
            raise ValueError("sideband must be 'usb' or 'lsb'")
        self.sideband = side
        self.agc_enabled = bool(agc_enabled)
        self.dc = DCBlocker(radius=dc_radius)
        self.agc_target = 10.0 ** (agc_target_dbfs / 20.0)
        self.agc_decay = agc_decay

    def stages(self, z) -> list:
        real = D.empty(z.numel(), "float32")
        N.call("iqa_real_part", N.ptr(z), c_int64(z.numel()), N.ptr(real), N.stream_ptr())
        chain = [("analytic", z), ("dc_block", self.dc.process(real))]
        if self.agc_enabled:
            chain.append(("agc", self.agc(chain[-1][1])))
        return chain

    def agc(self, audio, restarts=None):
        """gain += decay * (target / |s| - gain) for |s| > 1e-6, out = s * gain, gain = 1 at element 0 and at every index
        in ``restarts`` (device int64, optional) -- ssb.py:65-80 as a segmented scan."""
        n = int(audio.numel())
        if n == 0:
            return audio
        out = D.empty(n, "float32")
        count = 0 if restarts is None else int(restarts.numel())
        N.call("iqa_agc", N.ptr(audio), c_int64(n), c_double(self.agc_target), c_double(self.agc_decay),
               N.ptr(restarts) if count else c_void_p(0), c_int64(count), N.ptr(out), N.ptr(scan_workspace(n)), N.stream_ptr())
        return out

    def fused_params(self) -> N.DemodParams:
        return N.DemodParams(mode=N.DEMOD_MODE[self.sideband], agc_enabled=int(self.agc_enabled), deemph_alpha=0.0,
                             dc_radius=self.dc.radius, agc_target=self.agc_target, agc_decay=self.agc_decay)
