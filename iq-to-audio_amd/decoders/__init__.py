"""Pluggable demodulators on the HIP library (the reference's ``decoders`` package)."""
from __future__ import annotations

from .am import AMDecoder
from .base import Decoder, DecoderStats, GpuDecoder
from .nfm import NarrowbandFMDecoder
from .ssb import SSBDecoder

# mode string -> how to build its decoder from (deemph_us, agc_enabled); the AGC switch only reaches the SSB decoder
_BUILDERS = {
    "nfm": lambda deemph_us, agc: NarrowbandFMDecoder(deemph_us=deemph_us),
    "am": lambda deemph_us, agc: AMDecoder(),
    "usb": lambda deemph_us, agc: SSBDecoder(sideband="usb", agc_enabled=agc),
    "lsb": lambda deemph_us, agc: SSBDecoder(sideband="lsb", agc_enabled=agc),
}
_ALIASES = {"fm": "nfm", "ssb": "usb"}


def create_decoder(mode: str, *, deemph_us: float, agc_enabled: bool) -> Decoder:
    """The decoder for a ``--demod`` mode (reference decoders/__init__.py:9-24): nfm | fm, am, usb | ssb, lsb;
    anything else is a ``ValueError``."""
    key = mode.lower()
    build = _BUILDERS.get(_ALIASES.get(key, key))
    if build is None:
        raise ValueError(f"Unsupported demod mode '{key}'.")
    return build(deemph_us, agc_enabled)


__all__ = ["Decoder", "DecoderStats", "GpuDecoder", "create_decoder", "NarrowbandFMDecoder", "AMDecoder", "SSBDecoder"]
