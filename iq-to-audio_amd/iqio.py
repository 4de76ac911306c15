"""Capture ingest and audio egress without ffmpeg.

The reference decodes captures with an ffmpeg subprocess into float32 stereo
(``processing.py:107-279``) and encodes audio through a second ffmpeg
(``processing.py:381-524``).  Here the capture's *own* samples (int16 / uint8 / float32
interleaved I/Q) are handed to the GPU untouched -- the convert happens inside the
channelizer kernel -- so ingest is just a tolerant header parse plus a memory map.

Formats (reference ``input_formats.py:45-110``): WAV PCM_U8 / PCM_16 / FLOAT (incl. RF64 and
WAVE_FORMAT_EXTENSIBLE, data length ignored when it is 0 / 0xFFFFFFFF / larger than the file,
like ffmpeg's ``-ignore_length 1``) and raw ``.cu8`` / ``.cs16`` / ``.cf32`` / ``.iq``.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass
from pathlib import Path

import numpy as np

CODEC_TO_FMT = {"pcm_u8": "u8", "pcm_s16le": "s16", "pcm_f32le": "f32"}
FMT_TO_CODEC = {v: k for k, v in CODEC_TO_FMT.items()}
RAW_SUFFIX = {".cu8": "pcm_u8", ".cs16": "pcm_s16le", ".cf32": "pcm_f32le", ".iq": "pcm_s16le"}
FRAME_BYTES = {"u8": 2, "s16": 4, "f32": 8}
NP_DTYPE = {"u8": np.uint8, "s16": np.dtype("<i2"), "f32": np.dtype("<f4")}


@dataclass
class CaptureInfo:
    path: Path
    container: str  # "wav" | "raw"
    codec: str  # pcm_u8 | pcm_s16le | pcm_f32le
    fmt: str  # u8 | s16 | f32
    sample_rate: float | None
    data_offset: int
    n_frames: int

    @property
    def frame_bytes(self) -> int:
        return FRAME_BYTES[self.fmt]


def _parse_wav(path: Path) -> CaptureInfo:
    size = path.stat().st_size
    with path.open("rb") as fh:
        head = fh.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise ValueError(f"{path} is not a RIFF/RF64 WAVE file")
        rf64 = head[:4] == b"RF64"
        fmt_tag = channels = bits = None
        rate = None
        data_off = data_len = None
        ds64_len = None
        pos = 12
        while pos + 8 <= size:
            fh.seek(pos)
            hdr = fh.read(8)
            if len(hdr) < 8:
                break
            cid, clen = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            body = pos + 8
            if cid == b"ds64":
                raw = fh.read(min(clen, 28))
                if len(raw) >= 16:
                    ds64_len = struct.unpack("<Q", raw[8:16])[0]
            elif cid == b"fmt ":
                raw = fh.read(min(clen, 40))
                fmt_tag, channels, rate_i, _, _, bits = struct.unpack("<HHIIHH", raw[:16])
                rate = float(rate_i)
                if fmt_tag == 0xFFFE and len(raw) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                    fmt_tag = struct.unpack("<H", raw[24:26])[0]
            elif cid == b"data":
                data_off, data_len = body, clen
                break
            pos = body + clen + (clen & 1)
    if fmt_tag is None or data_off is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    if channels != 2:
        raise ValueError(f"{path}: expected 2 channels (I/Q), found {channels}")
    if fmt_tag == 1 and bits == 8:
        codec = "pcm_u8"
    elif fmt_tag == 1 and bits == 16:
        codec = "pcm_s16le"
    elif fmt_tag == 3 and bits == 32:
        codec = "pcm_f32le"
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format tag {fmt_tag}, {bits} bits)")
    avail = size - data_off
    if rf64 and ds64_len:
        data_len = ds64_len
    if data_len in (0, 0xFFFFFFFF) or data_len > avail:
        data_len = avail  # tolerate streaming writers that never patched the length
    fmt = CODEC_TO_FMT[codec]
    return CaptureInfo(path, "wav", codec, fmt, rate, data_off, data_len // FRAME_BYTES[fmt])


def probe_capture(path: Path, *, input_format: str | None = None, input_container: str | None = None,
                  input_sample_rate: float | None = None) -> CaptureInfo:
    """Resolve container/codec/sample-rate (reference input_formats.resolve_input_format
    :316-332 and probe.probe_sample_rate :32-37, without ffprobe/libsndfile)."""
    path = Path(path)
    suffix = path.suffix.lower()
    container = input_container
    if container is None:
        container = "raw" if suffix in RAW_SUFFIX else "wav"
    if container == "wav":
        info = _parse_wav(path)
        if input_format and input_format != info.codec:
            if input_format not in CODEC_TO_FMT:
                raise ValueError(f"Unsupported input format '{input_format}'")
            fmt = CODEC_TO_FMT[input_format]
            nbytes = info.n_frames * info.frame_bytes
            info = CaptureInfo(path, "wav", input_format, fmt, info.sample_rate, info.data_offset,
                               nbytes // FRAME_BYTES[fmt])
        if input_sample_rate is not None:
            info.sample_rate = float(input_sample_rate)
        return info
    if container != "raw":
        raise ValueError(f"Unsupported input container '{container}'")
    codec = input_format or RAW_SUFFIX.get(suffix)
    if codec not in CODEC_TO_FMT:
        raise ValueError(f"Cannot infer raw IQ format from suffix '{suffix}'; pass input_format")
    fmt = CODEC_TO_FMT[codec]
    return CaptureInfo(path, "raw", codec, fmt, float(input_sample_rate) if input_sample_rate else None, 0,
                       path.stat().st_size // FRAME_BYTES[fmt])


def map_frames(info: CaptureInfo) -> np.ndarray:
    """Memory map of the payload as a flat interleaved array (2*n_frames values)."""
    if info.n_frames == 0:
        return np.empty(0, dtype=NP_DTYPE[info.fmt])
    return np.memmap(info.path, dtype=NP_DTYPE[info.fmt], mode="r", offset=info.data_offset, shape=(2 * info.n_frames,))


_FILENAME_FREQ = re.compile(r"(?i)(\d+(?:\.\d+)?)([kmg]?)(?:hz)")
_UNIT = {"": 1.0, "k": 1e3, "m": 1e6, "g": 1e9}


def center_frequency_from_filename(path: Path) -> tuple[float | None, str]:
    """Largest ``<number>[k|M|G]Hz`` token >= 1 kHz in the file name (reference
    utils._center_frequency_from_filename :179-200).  Metadata-tag detection (ffprobe /
    libsndfile) is out of scope."""
    best = None
    for mt in _FILENAME_FREQ.finditer(Path(path).name):
        value = float(mt.group(1)) * _UNIT[mt.group(2).lower()]
        if value >= 1_000.0 and (best is None or value > best):
            best = value
    if best is None:
        return None, "unavailable"
    stem = Path(path).stem.lower()
    if stem.startswith("baseband_"):
        return best, "filename:sdrpp"
    if re.match(r"\d{2}-\d{2}-\d{2}_", stem):
        return best, "filename:sdrsharp"
    return best, "filename"


def write_wav_pcm16(path: Path, pcm: np.ndarray, sample_rate: int, channels: int = 1) -> None:
    """Minimal RIFF/WAVE PCM16 writer (the ``-acodec pcm_s16le`` leg of processing.py:399-418)."""
    data = np.ascontiguousarray(pcm, dtype="<i2").tobytes()
    block = 2 * channels
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, channels, int(sample_rate), int(sample_rate) * block, block, 16
    ) + b"data" + struct.pack("<I", len(data))
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("wb") as fh:
        fh.write(hdr)
        fh.write(data)


def write_wav_iq(path: Path, frames: np.ndarray, sample_rate: int, fmt: str = "s16") -> None:
    """Stereo I/Q WAV (PCM16 / U8 / float32) -- used by the synthetic benchmark generator."""
    arr = np.ascontiguousarray(frames).reshape(-1)
    if fmt == "s16":
        tag, bits, data = 1, 16, arr.astype("<i2").tobytes()
    elif fmt == "u8":
        tag, bits, data = 1, 8, arr.astype(np.uint8).tobytes()
    elif fmt == "f32":
        tag, bits, data = 3, 32, arr.astype("<f4").tobytes()
    else:
        raise ValueError(fmt)
    block = 2 * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, tag, 2, int(sample_rate), int(sample_rate) * block, block, bits
    ) + b"data" + struct.pack("<I", len(data) if len(data) < 2**32 else 0xFFFFFFFF)
    with Path(path).open("wb") as fh:
        fh.write(hdr)
        fh.write(data)


def encode_iq_slice(z: np.ndarray, codec: str, container: str = "raw") -> np.ndarray:
    """Interleaved I,Q values of a decimated slice in the INPUT's sample format (pass-through ``--demod none``).

    Headerless outputs follow the reference's own quantisers (processing.py:527-539): ``pcm_s16le`` = the value
    limited to [-1, 0.999969], times 32767, truncated toward zero; ``pcm_u8`` = the value limited to [-1, 1], mapped
    to (x + 1) * 127.5 and rounded half-to-even; ``pcm_f32le`` unchanged.
    WAV outputs go through libsndfile in the reference (``SoundFile.write`` of float32 frames, processing.py:557-584):
    third-party and absent here -- **parity unpinned**; restated from libsndfile's float->PCM rule with
    normalisation on: PCM_16 = rint(x * 32767), PCM_U8 = rint(x * 127) + 128 (saturated here; libsndfile wraps
    unless clipping is switched on), FLOAT unchanged.
    """
    pairs = np.ascontiguousarray(z, dtype=np.complex64).view(np.float32)  # I0, Q0, I1, Q1, ... as they lie in memory
    if codec == "pcm_f32le":
        return pairs.astype("<f4", copy=False)
    if codec not in ("pcm_s16le", "pcm_u8"):
        raise ValueError(f"Unsupported raw codec {codec}")
    if container == "wav":
        if codec == "pcm_s16le":
            return np.clip(np.rint(pairs.astype(np.float64) * 32767.0), -32768, 32767).astype("<i2")
        return np.clip(np.rint(pairs.astype(np.float64) * 127.0) + 128.0, 0, 255).astype(np.uint8)
    if codec == "pcm_s16le":
        return (np.minimum(np.maximum(pairs, np.float32(-1.0)), np.float32(0.999969)) * np.float32(32767.0)).astype("<i2")
    limited = np.minimum(np.maximum(pairs, np.float32(-1.0)), np.float32(1.0))
    return np.round((limited + np.float32(1.0)) * np.float32(127.5)).astype(np.uint8)


def read_wav_pcm16_mono(path: Path) -> tuple[np.ndarray, int]:
    """Read back a mono PCM16 WAV written by :func:`write_wav_pcm16` (tests)."""
    raw = Path(path).read_bytes()
    rate = struct.unpack("<I", raw[24:28])[0]
    n = struct.unpack("<I", raw[40:44])[0]
    return np.frombuffer(raw[44 : 44 + n], dtype="<i2"), rate
