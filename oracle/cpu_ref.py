"""oracle/cpu_ref.py -- CPU restatement of the iq-to-audio channelize->demodulate path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``iq-to-audio_amd/`` may import this
module; it is the *checker* used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path is the HIP library and it
raises when that library is missing -- it never falls back to this code.

Parity status: **pinned** for every stage up to the clipped float32 audio at the
channel rate.  ``oracle/gen_golden.py`` imports the reference's own stage classes
from ``/root/reference/src`` (in the build container only) and writes the fixtures
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against them bit-for-bit (float32 outputs) on every run, on any machine.
The 48 kHz resampler is **parity unpinned**: the reference delegates it to an
ffmpeg subprocess (libswresample, un-pinned, absent here); ``resample_48k`` below
is the build-defined specification of that stage.

Every function cites the reference lines it restates (paths relative to
``/root/reference/src/iq_to_audio/``).  The code is written from the behaviour
described in SURVEY.md section 8(a); it is a functional, explicit-state
formulation rather than the reference's class layout.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
from scipy import fft as _sfft
from scipy import signal as _ssig

_HERE = Path(__file__).resolve().parent

# --------------------------------------------------------------------------- #
# scalars / planning                                                          #
# --------------------------------------------------------------------------- #

MAX_CHUNK = 4_194_304


def tune_chunk_size(sample_rate: float, requested: int) -> int:
    """Effective chunk length.  processing.py:65-81."""
    base = requested if requested > 1 else 1
    if sample_rate <= 0:
        return base
    seconds = 0.25
    if sample_rate >= 2.0e6:
        seconds = 0.40
    if sample_rate >= 5.0e6:
        seconds = 0.50
    want = int(round(sample_rate * seconds))
    if want <= base:
        return base
    want = min(MAX_CHUNK, max(base, want))
    pow2 = 1 << math.ceil(math.log2(want))
    return int(min(max(pow2, base), MAX_CHUNK))


def decimation_for(sample_rate: float, fs_ch_target: float) -> tuple[int, float]:
    """Integer decimation + channel rate.  processing.py:885-890.

    Python ``round`` (half-to-even) first; if that leaves the channel rate more
    than 1.5x the target, fall back to ``floor``.
    """
    dec = max(1, int(round(sample_rate / fs_ch_target)))
    fs_ch = sample_rate / dec
    if fs_ch > fs_ch_target * 1.5:
        dec = max(int(math.floor(sample_rate / fs_ch_target)), 1)
        fs_ch = sample_rate / dec
    return dec, fs_ch


def channel_filter_params(sample_rate: float, bandwidth: float, decimation: int):
    """(num_taps, cutoff_hz, kaiser_beta).  processing.py:599-612."""
    guard = max(1000.0, 0.5 * bandwidth)
    cutoff = min(0.5 * bandwidth * 1.05, 0.9 * (sample_rate / (2.0 * max(decimation, 1))))
    if cutoff <= 0:
        raise ValueError("Invalid cutoff frequency for channel filter.")
    width = guard / sample_rate
    ntaps = int(np.clip(4.0 / max(width, 1e-8), 1024, 32768))
    if ntaps % 2 == 0:
        ntaps += 1
    return ntaps, cutoff, float(_ssig.kaiser_beta(80.0))


def design_channel_filter(sample_rate: float, bandwidth: float, decimation: int) -> np.ndarray:
    """Kaiser low-pass prototype, float64, unity DC gain.  processing.py:599-620."""
    ntaps, cutoff, beta = channel_filter_params(sample_rate, bandwidth, decimation)
    h = _ssig.firwin(ntaps, cutoff=cutoff, window=("kaiser", beta), fs=sample_rate)
    return np.asarray(h, dtype=np.float64)


# --------------------------------------------------------------------------- #
# ingest                                                                      #
# --------------------------------------------------------------------------- #

IQ_ORDERS = ("iq", "qi", "iq_inv", "qi_inv")


def ingest_to_complex64(raw: np.ndarray, fmt: str = "s16", iq_order: str = "iq") -> np.ndarray:
    """Interleaved capture -> complex64 block, as IQReader hands it to the DSP chain.

    processing.py:238-279 (+ ffmpeg's sample-format conversion, third-party:
    s16 -> x/32768, u8 -> (x-128)/128, f32 unchanged; restated from ffmpeg's
    documented behaviour, exact by construction in float32).
    ``raw`` is the flat interleaved array (I0,Q0,I1,Q1,...) or shape (n,2).
    """
    flat = np.asarray(raw).reshape(-1)
    if fmt == "s16":
        f = flat.astype(np.float32) / np.float32(32768.0)
    elif fmt == "u8":
        f = (flat.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif fmt == "f32":
        f = flat.astype(np.float32, copy=False)
    else:
        raise ValueError(f"unknown ingest format {fmt!r}")
    if iq_order not in IQ_ORDERS:
        raise ValueError(f"Unsupported iq_order '{iq_order}'")
    even, odd = f[0::2], f[1::2]
    i, q = (even, odd) if iq_order.startswith("iq") else (odd, even)
    if iq_order.endswith("_inv"):
        q = -q
    out = np.empty(i.size, dtype=np.complex64)
    out.real = i
    out.imag = q
    return out


# --------------------------------------------------------------------------- #
# streaming stages with explicit state                                        #
# --------------------------------------------------------------------------- #


@dataclass
class NcoState:
    """processing.py:282-287: phase starts at 0, increment = -2*pi*f_off/fs."""

    freq_offset: float
    sample_rate: float
    phase: float = 0.0

    @property
    def increment(self) -> float:
        return -2.0 * np.pi * self.freq_offset / self.sample_rate


def nco_mix(x: np.ndarray, st: NcoState, sign: int) -> np.ndarray:
    """Frequency translation.  processing.py:289-297.

    float64 phase ramp ``phase + sign*inc*n``; oscillator rounded to complex64;
    complex64 product; the carried phase is wrapped mod 2*pi once per call.
    """
    if x.size == 0:
        return x
    k = np.arange(x.size, dtype=np.float64)
    step = sign * st.increment
    ramp = st.phase + step * k
    osc = np.exp(1j * ramp).astype(np.complex64)
    st.phase = (st.phase + step * x.size) % (2.0 * np.pi)
    return np.asarray(x.astype(np.complex64, copy=False) * osc, dtype=np.complex64)


@dataclass
class OverlapSaveState:
    """processing.py:303-323: F = next_pow2(B + L - 1), H = fft(zero-padded taps)."""

    taps: np.ndarray
    block: int
    hist: np.ndarray = field(init=False)
    nfft: int = field(init=False)
    spectrum: np.ndarray = field(init=False)

    def __post_init__(self) -> None:
        if self.block <= 0:
            raise ValueError("block_size must be positive")
        ntaps = len(self.taps)
        self.nfft = 1 << math.ceil(math.log2(self.block + ntaps - 1))
        padded = np.zeros(self.nfft, dtype=np.complex128)
        padded[:ntaps] = np.asarray(self.taps).astype(np.complex128)
        self.spectrum = _sfft.fft(padded)
        self.hist = np.zeros(ntaps - 1, dtype=np.complex64)

    @property
    def ntaps(self) -> int:
        return len(self.taps)


def overlap_save(x: np.ndarray, st: OverlapSaveState) -> np.ndarray:
    """Causal linear convolution by FFT overlap-save.  processing.py:325-346.

    Per segment of <= block samples: [history(L-1) | segment] as complex128,
    zero-padded to F, ifft(fft(.)*H), keep [L-1 : L-1+len(segment)], round to
    complex64.  History = last L-1 *input* samples (short segments shift in).
    """
    if x.size == 0:
        return x
    keep = st.ntaps - 1
    src = x.astype(np.complex64)
    pieces = []
    pos = 0
    while pos < src.size:
        seg = src[pos : pos + st.block]
        pos += seg.size
        frame = np.zeros(st.nfft, dtype=np.complex128)
        frame[:keep] = st.hist
        frame[keep : keep + seg.size] = seg
        y = _sfft.ifft(_sfft.fft(frame) * st.spectrum)
        pieces.append(y[keep : keep + seg.size].astype(np.complex64))
        if keep:
            if seg.size >= keep:
                st.hist = seg[-keep:].copy()
            else:
                st.hist = np.concatenate([st.hist[seg.size :], seg]).astype(np.complex64)
    return np.concatenate(pieces)


@dataclass
class DecimState:
    """processing.py:349-352."""

    factor: int
    offset: int = 0


def decimate(x: np.ndarray, st: DecimState) -> np.ndarray:
    """Keep global indices 0, D, 2D, ...  processing.py:354-360."""
    d = max(1, st.factor)
    if d == 1 or x.size == 0:
        return x
    first = (-st.offset) % d
    out = x[first::d]
    st.offset = (st.offset + x.size) % d
    return out


def choose_mix_sign(
    warmup: np.ndarray, sample_rate: float, freq_offset: float, taps: np.ndarray, decimation: int
) -> int:
    """Mixer-sign probe on the first chunk.  processing.py:623-663."""
    if warmup.size == 0:
        return 1
    ntaps = len(taps)
    limit = max(int(sample_rate * 0.05), ntaps * 4, 131_072)
    take = min(warmup.size, limit)
    if take < ntaps:
        take = min(warmup.size, ntaps * 2)
    snip = warmup[:take].astype(np.complex64, copy=False)
    idx = np.arange(snip.size, dtype=np.float64)
    d = max(decimation, 1)
    blk = min(snip.size, max(ntaps, 16_384))
    winner, best = 1, -np.inf
    for sign in (1, -1):
        lo = np.exp(-1j * sign * 2.0 * np.pi * freq_offset * idx / sample_rate).astype(
            np.complex64, copy=False
        )
        y = overlap_save(snip * lo, OverlapSaveState(taps, blk))[::d]
        if y.size == 0:
            p = -np.inf
        else:
            skip = min(ntaps, y.size // 4)
            tail = y[skip:]
            if tail.size == 0:
                tail = y
            p = float(np.mean(np.abs(tail) ** 2))
        if p > best:
            best, winner = p, sign
    return winner


# --------------------------------------------------------------------------- #
# sequential float32 recurrences (C fast path + pure-Python statement)        #
# --------------------------------------------------------------------------- #

_SEQ_LIB = None


def build_c(force: bool = False) -> Path:
    """Compile oracle/seq_f32.c -> oracle/_build/liboracle_seq.so (gcc)."""
    so = _HERE / "_build" / "liboracle_seq.so"
    src = _HERE / "seq_f32.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return so


def _seq_lib():
    global _SEQ_LIB
    if _SEQ_LIB is None:
        lib = ctypes.CDLL(str(build_c()))
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.dc_block_f32.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_double, dp, dp]
        lib.dc_block_f32.restype = None
        lib.agc_f32.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_double, ctypes.c_double]
        lib.agc_f32.restype = None
        lib.dc_block_f64.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_double, dp, dp]
        lib.dc_block_f64.restype = None
        lib.agc_f64.argtypes = [fp, fp, ctypes.c_size_t, ctypes.c_double, ctypes.c_double]
        lib.agc_f64.restype = None
        _SEQ_LIB = lib
    return _SEQ_LIB


@dataclass
class DcState:
    """decoders/common.py:9-14."""

    radius: float = 0.995
    x_prev: float = 0.0
    y_prev: float = 0.0


def dc_block_py(x: np.ndarray, st: DcState) -> np.ndarray:
    """Pure-Python statement of decoders/common.py:16-30 (slow; small cases)."""
    if x.size == 0:
        return x
    out = np.empty(x.size, dtype=np.float32)
    xp, yp, r = st.x_prev, st.y_prev, st.radius
    for i, s in enumerate(x.astype(np.float32, copy=False)):
        y = s - xp + r * yp
        out[i] = y
        xp, yp = s, y
    st.x_prev, st.y_prev = float(xp), float(yp)
    return out


def dc_block(x: np.ndarray, st: DcState) -> np.ndarray:
    """y[n] = x[n] - x[n-1] + r*y[n-1], float32 sequential.  decoders/common.py:16-30."""
    if x.size == 0:
        return x
    xin = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(xin.size, dtype=np.float32)
    xp, yp = ctypes.c_double(st.x_prev), ctypes.c_double(st.y_prev)
    fp = ctypes.POINTER(ctypes.c_float)
    _seq_lib().dc_block_f32(
        xin.ctypes.data_as(fp), out.ctypes.data_as(fp), xin.size, st.radius,
        ctypes.byref(xp), ctypes.byref(yp),
    )
    st.x_prev, st.y_prev = float(xp.value), float(yp.value)
    return out


AGC_TARGET = 10.0 ** (-12.0 / 20.0)
AGC_DECAY = 0.001


def agc_py(x: np.ndarray, target: float = AGC_TARGET, decay: float = AGC_DECAY) -> np.ndarray:
    """Pure-Python statement of decoders/ssb.py:65-80 (slow; small cases)."""
    if x.size == 0:
        return x
    g = 1.0
    out = np.empty(x.size, dtype=np.float32)
    for i, s in enumerate(x):
        m = abs(s)
        if m > 1e-6:
            g += decay * (target / m - g)
        out[i] = s * g
    return out


def agc(x: np.ndarray, target: float = AGC_TARGET, decay: float = AGC_DECAY) -> np.ndarray:
    """Per-call AGC (gain restarts at 1.0 every call).  decoders/ssb.py:65-80."""
    if x.size == 0:
        return x
    xin = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(xin.size, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _seq_lib().agc_f32(xin.ctypes.data_as(fp), out.ctypes.data_as(fp), xin.size, target, decay)
    return out


def ssb_demod_f64(z: np.ndarray, st: "DcState", *, lsb: bool = False, agc_enabled: bool = True) -> np.ndarray:
    """decoders/ssb.py:39-61 with both recurrences evaluated in float64 (oracle/seq_f32.c: dc_block_f64, agc_f64).
    NOT the reference's arithmetic: the comparison target that separates logic from rounding in the SSB+AGC tests.
    ``st`` carries the DC blocker's state across calls as doubles; the gain restarts at 1.0 every call."""
    if z.size == 0:
        return np.empty(0, dtype=np.float32)
    base = np.ascontiguousarray((np.conj(z) if lsb else z).real, dtype=np.float32)
    dc = np.empty(base.size, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    xp, yp = ctypes.c_double(st.x_prev), ctypes.c_double(st.y_prev)
    _seq_lib().dc_block_f64(base.ctypes.data_as(fp), dc.ctypes.data_as(fp), base.size, st.radius, ctypes.byref(xp), ctypes.byref(yp))
    st.x_prev, st.y_prev = float(xp.value), float(yp.value)
    if not agc_enabled:
        return dc
    out = np.empty(dc.size, dtype=np.float32)
    _seq_lib().agc_f64(dc.ctypes.data_as(fp), out.ctypes.data_as(fp), dc.size, AGC_TARGET, AGC_DECAY)
    return out


# --------------------------------------------------------------------------- #
# demodulators                                                                #
# --------------------------------------------------------------------------- #


@dataclass
class QuadState:
    """decoders/nfm.py:14-15: previous sample starts at 1+0j."""

    prev: np.complex64 = np.complex64(1 + 0j)


def quadrature(z: np.ndarray, st: QuadState) -> np.ndarray:
    """angle(z[n]*conj(z[n-1])) in radians/sample, float32.  decoders/nfm.py:17-24."""
    if z.size == 0:
        return np.empty(0, dtype=np.float32)
    lag = np.concatenate(([st.prev], z[:-1]))
    out = np.asarray(np.angle(z * np.conj(lag)), dtype=np.float32)
    st.prev = z[-1]
    return out


def deemph_alpha(deemph_us: float, fs_ch: float) -> float:
    """decoders/nfm.py:39-41."""
    tau = max(deemph_us * 1e-6, 1e-6)
    return math.exp(-1.0 / (fs_ch * tau))


@dataclass
class DeemphState:
    alpha: float
    state: float = 0.0  # lfilter zi/zf (= alpha * y[last]), carried as a Python float


def deemphasis(x: np.ndarray, st: DeemphState) -> np.ndarray:
    """y[n] = (1-a) x[n] + a y[n-1] via float64 lfilter.  decoders/nfm.py:48-62."""
    if x.size == 0:
        return x
    b = np.array([1.0 - st.alpha], dtype=np.float64)
    a = np.array([1.0, -st.alpha], dtype=np.float64)
    y, zf = _ssig.lfilter(b, a, x.astype(np.float32, copy=False), zi=np.array([st.state]))
    st.state = float(zf[0])
    return np.asarray(y, dtype=np.float32)


def _rms_dbfs(y: np.ndarray) -> float:
    """decoders/nfm.py:88-89 (same in am.py:31-32, ssb.py:46-47)."""
    rms = math.sqrt(float(np.mean(y.astype(np.float64) ** 2)) + 1e-18)
    return 20.0 * math.log10(rms + 1e-12)


@dataclass
class DemodState:
    """One decoder instance (create_decoder, decoders/__init__.py:9-24)."""

    mode: str
    fs_ch: float
    deemph_us: float = 300.0
    agc_enabled: bool = True
    quad: QuadState = field(default_factory=QuadState)
    deemph: DeemphState | None = None
    dc: DcState = field(default_factory=DcState)

    def __post_init__(self) -> None:
        m = self.mode.lower()
        if m in ("nfm", "fm"):
            self.kind = "nfm"
            self.deemph = DeemphState(deemph_alpha(self.deemph_us, self.fs_ch))
        elif m == "am":
            self.kind = "am"
        elif m in ("usb", "ssb", "lsb"):
            self.kind = "ssb"
        else:
            raise ValueError(f"Unsupported demod mode '{m}'.")


def demodulate(z: np.ndarray, st: DemodState) -> tuple[np.ndarray, float]:
    """decoder.process(): (audio float32 at fs_ch, rms dBFS).

    nfm: decoders/nfm.py:82-97 -- discriminator then de-emphasis; agc ignored.
    am : decoders/am.py:25-40  -- |z| then DC blocker; agc ignored.
    ssb: decoders/ssb.py:39-61 -- real(z) (usb) / real(conj z) (lsb, identical),
         DC blocker, then per-call AGC when enabled.
    """
    if st.kind == "nfm":
        y = deemphasis(quadrature(z, st.quad), st.deemph)
    elif st.kind == "am":
        y = dc_block(np.abs(z).astype(np.float32, copy=False), st.dc)
    else:
        base = (np.conj(z) if st.mode.lower() == "lsb" else z).real.astype(np.float32, copy=False)
        y = dc_block(base, st.dc)
        if st.agc_enabled:
            y = agc(y)
    return y, (_rms_dbfs(y) if y.size else -240.0)


def writer_clip(audio: np.ndarray, peak: float) -> tuple[np.ndarray, float]:
    """AudioWriter.write: pre-clip peak, clip to +-0.99 float32.  processing.py:440-456."""
    if audio.size == 0:
        return audio, peak
    peak = max(peak, float(np.max(np.abs(audio))))
    return np.clip(audio, -0.99, 0.99).astype(np.float32, copy=False), peak


# --------------------------------------------------------------------------- #
# whole chain                                                                 #
# --------------------------------------------------------------------------- #


@dataclass
class ChainResult:
    audio: np.ndarray  # clipped float32 at fs_channel (the parity artefact)
    decimated: np.ndarray  # complex64 at fs_channel
    mix_sign: int
    decimation: int
    fs_channel: float
    chunk: int
    ntaps: int
    audio_peak: float
    rms_dbfs: list


def run_chain(
    raw: np.ndarray,
    *,
    sample_rate: float,
    freq_offset: float,
    bandwidth: float = 12_500.0,
    demod_mode: str = "nfm",
    fs_ch_target: float = 96_000.0,
    deemph_us: float = 300.0,
    agc_enabled: bool = True,
    chunk_size: int = 1_048_576,
    filter_block: int = 65_536,
    iq_order: str = "iq",
    mix_sign_override: int | None = None,
    fmt: str = "s16",
    tune_chunk: bool = True,
    keep_decimated: bool = True,
) -> ChainResult:
    """The hand-composed chain of SURVEY.md section 8(c) (= ProcessingPipeline.run's
    per-chunk loop, processing.py:1070-1154, minus the two ffmpeg processes):

        ingest -> nco_mix -> overlap_save -> decimate -> demodulate -> writer_clip

    fed chunk by chunk with C = tune_chunk_size(fs, chunk_size); the first chunk is
    also the warm-up snippet for choose_mix_sign unless the sign is overridden.
    """
    dec, fs_ch = decimation_for(sample_rate, fs_ch_target)
    chunk = tune_chunk_size(sample_rate, chunk_size) if tune_chunk else chunk_size
    taps = design_channel_filter(sample_rate, bandwidth, dec)
    nco = NcoState(freq_offset, sample_rate)
    fir = OverlapSaveState(taps, filter_block)
    dst = DecimState(dec)
    dem = DemodState(demod_mode, fs_ch, deemph_us=deemph_us, agc_enabled=agc_enabled)
    flat = np.asarray(raw).reshape(-1)
    nsamp = flat.size // 2
    sign = None
    peak = 0.0
    aud, decs, stats = [], [], []
    for start in range(0, nsamp, chunk):
        blk = ingest_to_complex64(flat[2 * start : 2 * min(start + chunk, nsamp)], fmt, iq_order)
        if sign is None:
            sign = (
                mix_sign_override
                if mix_sign_override in (1, -1)
                else choose_mix_sign(blk, sample_rate, freq_offset, taps, dec)
            )
        z = decimate(overlap_save(nco_mix(blk, nco, sign), fir), dst)
        if keep_decimated:
            decs.append(np.array(z, copy=True))
        y, db = demodulate(z, dem)
        y, peak = writer_clip(y, peak)
        aud.append(y)
        stats.append(db)
    return ChainResult(
        audio=np.concatenate(aud) if aud else np.empty(0, np.float32),
        decimated=np.concatenate(decs) if decs else np.empty(0, np.complex64),
        mix_sign=sign if sign is not None else 1,
        decimation=dec,
        fs_channel=fs_ch,
        chunk=chunk,
        ntaps=len(taps),
        audio_peak=peak,
        rms_dbfs=stats,
    )


# --------------------------------------------------------------------------- #
# synthetic capture (the --benchmark input definition)                        #
# --------------------------------------------------------------------------- #


def synth_capture_s16(
    sample_rate: float,
    seconds: float,
    freq_offset: float,
    *,
    amplitude: float = 0.7,
    noise_std: float = 0.02,
    seed: int = 42,
) -> np.ndarray:
    """int16 interleaved I/Q of the reference's synthetic benchmark capture.

    benchmark.py:19-38: tone exp(j 2 pi f_off t) * 0.7 + N(0, 0.02) per rail from
    default_rng(42).normal(size=(N,2)); float32; clip +-0.999; libsndfile PCM_16
    write = lrint(x*32767) (third-party, restated).
    Returns shape (N, 2) int16.
    """
    n = int(round(sample_rate * seconds))
    if n <= 0:
        raise ValueError("Benchmark duration is too short to generate samples.")
    t = np.arange(n, dtype=np.float64) / sample_rate
    tone = np.exp(1j * 2.0 * math.pi * freq_offset * t)
    noise = np.random.default_rng(seed).normal(scale=noise_std, size=(n, 2))
    iq = np.column_stack((amplitude * tone.real + noise[:, 0], amplitude * tone.imag + noise[:, 1]))
    iq = np.clip(iq.astype(np.float32), -0.999, 0.999)
    return np.rint(iq.astype(np.float64) * 32767.0).astype(np.int16)


# --------------------------------------------------------------------------- #
# 48 kHz resampler -- BUILD-DEFINED SPEC (parity unpinned, see module docstring)
# --------------------------------------------------------------------------- #

RS_ZERO_CROSSINGS = 16
RS_CUTOFF = 0.97
RS_KAISER_BETA = 9.0
RS_OUT_RATE = 48_000


def resampler_plan(fs_channel: float, out_rate: int = RS_OUT_RATE):
    """(in_rate_int, up, down): the integer rate the reference declares to ffmpeg
    (processing.py:389-397, round(fs_ch)) reduced against 48 000."""
    rin = max(1, int(round(fs_channel)))
    g = math.gcd(out_rate, rin)
    return rin, out_rate // g, rin // g


def resampler_prototype(up: int, down: int) -> np.ndarray:
    """Float64 Kaiser-windowed-sinc prototype at the common rate up*in_rate.

    h[i], i in [-half, half], half = Z*max(up,down):
        h[i] = sinc(fc*i/M) * I0(beta*sqrt(1-(i/half)^2)) / I0(beta),  M = max(up,down), fc = 0.97
    scaled so that sum(h) == up (unity pass-band gain after zero-stuffing).
    Z = 16, beta = 9, fc = 0.97 mirror libswresample's defaults (filter_size 32,
    kaiser_beta 9, cutoff 0.97) but this is our own definition, not a port.
    """
    m = max(up, down)
    half = RS_ZERO_CROSSINGS * m
    i = np.arange(-half, half + 1, dtype=np.float64)
    u = i / half
    win = np.i0(RS_KAISER_BETA * np.sqrt(np.clip(1.0 - u * u, 0.0, 1.0))) / np.i0(RS_KAISER_BETA)
    h = np.sinc(RS_CUTOFF * i / m) * win
    return h * (up / h.sum())


def resample_48k(audio: np.ndarray, fs_channel: float, out_rate: int = RS_OUT_RATE) -> np.ndarray:
    """Zero-phase rational polyphase resample of the clipped float32 audio.

        y[j] = sum_n x[n] * h[j*down - n*up],   j in [0, ceil(n*up/down)),  x = 0 outside [0, n)

    accumulated in float64, returned float32.  Implemented with scipy's upfirdn
    (a polyphase evaluator of exactly this sum) on the float64 prototype.
    """
    _, up, down = resampler_plan(fs_channel, out_rate)
    x = np.asarray(audio, dtype=np.float64)
    n_out = -(-x.size * up // down)
    if x.size == 0:
        return np.empty(0, dtype=np.float32)
    if up == 1 and down == 1:
        return x.astype(np.float32)
    h = resampler_prototype(up, down)
    half = (h.size - 1) // 2
    pad = (-half) % down  # leading zeros so that (half + pad) is a multiple of down
    hp = np.concatenate([np.zeros(pad), h])
    y = _ssig.upfirdn(hp, x, up, down)
    first = (half + pad) // down
    y = y[first : first + n_out]
    if y.size < n_out:
        y = np.concatenate([y, np.zeros(n_out - y.size)])
    return y.astype(np.float32)


def float_to_pcm16(y: np.ndarray) -> np.ndarray:
    """Build-defined 48 kHz quantiser: round-half-even of y*32768, saturated.
    (The reference's is libswresample's float->s16 inside ffmpeg; unpinned.)"""
    return np.clip(np.rint(y.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


# --------------------------------------------------------------------------- #
# spectrum / waterfall -- parity PINNED by tests/golden/spectrum.npz,   #
# generated from the reference's own functions by oracle/gen_golden_spectrum.py            #
# --------------------------------------------------------------------------- #

PSD_FLOOR = 1e-18


def psd_frame_db(frame: np.ndarray, nfft: int, sample_rate: float) -> np.ndarray:
    """One fftshift-ed PSD frame in dB.  spectrum.py:29-44 (compute_psd) == :156-171 (_SlidingFFT.psd):
    Hann window of the frame's own length n, complex128 FFT zero-padded to nfft,
    |X|^2 / (n * fs * mean(w^2) + 1e-18), 10*log10(. + 1e-18)."""
    n = frame.size
    w = np.hanning(n).astype(np.float64)
    scale = n * sample_rate * (np.sum(w * w) / n) + PSD_FLOOR
    spec = np.fft.fftshift(_sfft.fft(frame.astype(np.complex128) * w, n=nfft))
    power = (spec.real * spec.real + spec.imag * spec.imag) / scale
    return (10.0 * np.log10(np.abs(power) + PSD_FLOOR)).astype(np.float64)


def psd_freqs(nfft: int, sample_rate: float) -> np.ndarray:
    """spectrum.py:38-41, :152."""
    return np.fft.fftshift(np.fft.fftfreq(nfft, d=1.0 / sample_rate)).astype(np.float64)


def compute_psd(samples: np.ndarray, sample_rate: float, nfft: int = 1 << 18):
    """spectrum.py:15-45: PSD of the first nfft samples (all of them, zero-padded, if fewer)."""
    x = np.asarray(samples)
    if x.size == 0:
        raise ValueError("Cannot compute PSD for an empty signal.")
    return psd_freqs(nfft, sample_rate), psd_frame_db(x[:nfft], nfft, sample_rate)


def sliding_window_starts(block_sizes, nfft: int, hop: int):
    """Every window the reference's iterator yields (spectrum.py:96-130) for blocks of the given sizes: windows of
    nfft every hop over (carry + block), the unconsumed tail carried to the next block, no trailing partial window.
    Returns [(reported_start, true_start)]: ``true_start`` is the window's position in the concatenated stream;
    ``reported_start`` is the index the reference hands on (its ``offset`` is decremented by the carry length
    although it already points at the carry, :110-111 with :125-126, so time stamps after a block boundary lag
    by that many samples) -- reproduced because the waterfall's ``times`` are derived from it."""
    out, carry, offset, consumed = [], 0, 0, 0
    for size in block_sizes:
        if size == 0:
            continue
        total = carry + size
        base = consumed - carry  # true stream position of (carry + block)[0]
        consumed += size
        offset -= carry
        if total < nfft:
            carry = total
            offset += total
            continue
        start = 0
        while start + nfft <= total:
            out.append((offset + start, base + start))
            start += hop
        tail = max(total - start, 0)
        offset += total - tail
        carry = min(tail, nfft)
    return out


def waterfall_reduce(slices: list, times: list, max_slices: int):
    """spectrum.py:190-208: while more than max_slices rows, average neighbours pairwise in float64 (rounded back
    to float32), keep an odd last row, keep the first time stamp of every pair."""
    while len(slices) > max_slices:
        nxt, nt = [], []
        for i in range(0, len(slices), 2):
            if i + 1 < len(slices):
                nxt.append(((slices[i].astype(np.float64) + slices[i + 1].astype(np.float64)) / 2.0).astype(np.float32))
            else:
                nxt.append(slices[i])
            nt.append(times[i])
        slices, times = nxt, nt
    return slices, times


def streaming_waterfall(chunks, sample_rate: float, *, nfft: int, hop: int | None = None, max_slices: int = 400):
    """spectrum.py:58-93: (freqs, mean of the frames' dB values, times float32, matrix float32 [slices][nfft], frames)."""
    hop = max(1, hop or nfft // 4)
    max_slices = max(1, int(max_slices))
    blocks = [np.asarray(c, dtype=np.complex64).reshape(-1) for c in chunks if c is not None]
    blocks = [b for b in blocks if b.size]
    stream = np.concatenate(blocks) if blocks else np.empty(0, dtype=np.complex64)
    starts = sliding_window_starts([b.size for b in blocks], nfft, hop)
    if not starts:
        raise ValueError("Input did not contain enough samples for one FFT frame.")
    total = np.zeros(nfft, dtype=np.float64)
    slices, times = [], []
    for reported, s0 in starts:
        db = psd_frame_db(stream[s0 : s0 + nfft], nfft, sample_rate)
        total += db
        slices.append(db.astype(np.float32))
        times.append(float(reported / sample_rate))
        slices, times = waterfall_reduce(slices, times, max_slices)
    return (psd_freqs(nfft, sample_rate), total / len(starts), np.asarray(times, dtype=np.float32),
            np.stack(slices, axis=0).astype(np.float32), len(starts))


def fft_workers_auto() -> int | None:
    """processing.py:697-732 (non-frozen branch): min(12, max(2, cores-1)); None if <=2 cores."""
    n = os.cpu_count() or 1
    return None if n <= 2 else min(12, max(2, n - 1))
