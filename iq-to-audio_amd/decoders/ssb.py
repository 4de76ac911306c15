"""SSB: real part + DC blocker + per-call AGC (reference decoders/ssb.py), on the HIP library.

As in the reference, USB and LSB produce identical audio (real(conj z) == real(z)) and the
AGC gain restarts at 1.0 on every ``process()`` call.
"""
from __future__ import annotations

from ctypes import c_double, c_int64, c_void_p

from .. import _dev as D
from .. import _native as N
from .base import Decoder, DecoderStats
from .common import DCBlocker, rms_dbfs_of, scan_workspace


class SSBDecoder(Decoder):
    name = "ssb"

    def __init__(self, sideband: str, agc_enabled: bool, dc_radius: float = 0.995, agc_target_dbfs: float = -12.0,
                 agc_decay: float = 0.001):
        sideband = sideband.lower()
        if sideband not in {"usb", "lsb"}:
            raise ValueError("sideband must be 'usb' or 'lsb'")
        self._sideband = sideband
        self._agc_enabled = agc_enabled
        self._dc_blocker = DCBlocker(radius=dc_radius)
        self._last_stats = None
        self._intermediates = {}
        self._sample_rate = 0.0
        self._agc_level = 10.0 ** (agc_target_dbfs / 20.0)
        self._agc_decay = agc_decay

    def setup(self, sample_rate: float) -> None:
        self._sample_rate = sample_rate

    def process(self, samples, *, agc_restarts=None):
        """``agc_restarts`` (device int64 tensor, optional) lists extra element indices where
        the AGC gain restarts -- used by the pipeline to run many reference chunks in one call."""
        if self._sample_rate == 0.0:
            raise RuntimeError("Decoder.setup(sample_rate) must be called before processing data.")
        z = D.to_device(samples, "complex64")
        baseband = D.empty(z.numel(), "float32")
        N.call("iqa_real_part", N.ptr(z), c_int64(z.numel()), N.ptr(baseband), N.stream_ptr())
        dc_audio = self._dc_blocker.process(baseband)
        processed = self._apply_agc(dc_audio, agc_restarts) if self._agc_enabled else dc_audio
        stats = DecoderStats(rms_dbfs=rms_dbfs_of(processed))
        self._last_stats = stats
        if z.numel():
            inter = {"analytic": (z, self._sample_rate), "dc_block": (dc_audio, self._sample_rate)}
            if self._agc_enabled:
                inter["agc"] = (processed, self._sample_rate)
            inter["audio"] = (processed, self._sample_rate)
            self._intermediates = inter
        return D.like_input(processed, samples), stats

    def _apply_agc(self, audio, restarts=None):
        n = int(audio.numel())
        if n == 0:
            return audio
        out = D.empty(n, "float32")
        work = scan_workspace(n)
        nres = 0 if restarts is None else int(restarts.numel())
        N.call("iqa_agc", N.ptr(audio), c_int64(n), c_double(self._agc_level), c_double(self._agc_decay),
               N.ptr(restarts) if nres else c_void_p(0), c_int64(nres), N.ptr(out), N.ptr(work), N.stream_ptr())
        return out

    def finalize(self) -> None:
        return

    def intermediates(self) -> dict:
        return {k: (v.cpu().numpy().copy(), r) for k, (v, r) in self._intermediates.items()}

    @property
    def last_stats(self):
        return self._last_stats
