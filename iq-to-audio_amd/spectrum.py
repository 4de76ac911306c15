"""GPU stand-in for the reference's ``src/iq_to_audio/spectrum.py``.

Same functions, arguments and return types as the reference -- :func:`compute_psd` (spectrum.py:15-45),
:func:`streaming_waterfall` (:58-93) with :class:`WaterfallResult` (:48-52) -- computed by
``iqa_psd_frames`` (window, rocFFT double-complex FFT, |X|^2/scale, 10 log10, fftshift, for a batch of frames per
call) and ``iqa_pair_average_rows`` (the waterfall's pairwise reduction).  The sliding-window bookkeeping
(``_sliding_windows``, :96-130) and the aggregator's slice/time lists (:174-208) stay on the host; the slices
themselves stay on the device until ``finalize``.  ``fft_workers`` is accepted and ignored (it selects SciPy
threads in the reference).  No CPU path: without a GPU or the built library the calls raise ``RuntimeError``.
"""
from __future__ import annotations

import logging
from collections.abc import Iterable
from ctypes import c_double, c_int32, c_int64
from dataclasses import dataclass

import numpy as np

from . import _dev as D
from . import _native as N
from . import dsp_plan as P

LOG = logging.getLogger(__name__)

_NUMPY_EPS = 1e-18
#: frames per ``iqa_psd_frames`` call (64 x 2^18 double-complex points = 268 MB of work space)
BATCH_FRAMES = 64


def _freqs(nfft: int, sample_rate: float) -> np.ndarray:
    return np.fft.fftshift(np.fft.fftfreq(nfft, d=1.0 / sample_rate)).astype(np.float64)


class _PsdEngine:
    """Window, scale and work space for one (window length, nfft, sample rate) -- ``_SlidingFFT`` (spectrum.py:143-171)."""

    def __init__(self, *, sample_rate: float, nfft: int, use: int | None = None, fmt: str = "f32", iq_order: str = "iq"):
        if iq_order not in N.ORDER:
            raise ValueError(f"Unsupported iq_order '{iq_order}'")
        self.sample_rate, self.nfft, self.use = float(sample_rate), int(nfft), int(use if use is not None else nfft)
        self.fmt, self.iq_order = fmt, iq_order
        window = np.hanning(self.use).astype(np.float64)
        self.win_power = float(np.sum(window**2) / self.use)
        self.scale = (self.use * self.sample_rate * self.win_power) + _NUMPY_EPS
        self.window_dev = D.from_numpy(window)
        self._work = None

    def frames(self, samples_dev, n_samples: int, first: int, hop: int, n_frames: int, *, want_f64: bool, want_f32: bool,
               sum_db=None):
        """PSD (dB, fftshift-ed) of ``n_frames`` frames starting at ``first`` with stride ``hop``."""
        if self._work is None or self._work.numel() < 2 * n_frames * self.nfft:
            self._work = D.empty(2 * max(n_frames, 1) * self.nfft, "float64")
        out64 = D.empty(n_frames * self.nfft, "float64") if want_f64 else None
        out32 = D.empty(n_frames * self.nfft, "float32") if want_f32 else None
        N.call("iqa_psd_frames", c_int32(P.FMT_CODE[self.fmt]), c_int32(N.ORDER[self.iq_order]), N.ptr(samples_dev),
               c_int64(n_samples), c_int64(first), c_int64(hop), c_int32(n_frames), c_int32(self.nfft), c_int32(self.use),
               N.ptr(self.window_dev), c_double(self.scale), N.ptr(self._work), N.ptr(out64), N.ptr(out32), N.ptr(sum_db),
               N.stream_ptr())
        return out64, out32


def _as_complex_dev(samples):
    if D.is_tensor(samples):
        return D.to_device(samples, "complex64")
    return D.to_device(np.asarray(samples, dtype=np.complex64), "complex64")


def compute_psd(samples, sample_rate: float, nfft: int = 1 << 18, *, fft_workers: int | None = None):
    """Single-sided PSD (dBFS) of complex samples -- reference spectrum.py:15-45: the first ``nfft`` samples (all of
    them, zero-padded, when there are fewer), Hann window of that length, float64 FFT, ``|X|^2/(n fs P_w)`` in dB,
    fftshift-ed.  Returns ``(freqs, psd_db)`` as float64 NumPy arrays."""
    n = int(samples.numel()) if D.is_tensor(samples) else int(np.asarray(samples).size)
    if n == 0:
        raise ValueError("Cannot compute PSD for an empty signal.")
    use = min(n, nfft)
    x = _as_complex_dev(samples)
    eng = _PsdEngine(sample_rate=sample_rate, nfft=nfft, use=use)
    out64, _ = eng.frames(x, n, 0, 1, 1, want_f64=True, want_f32=False)
    return _freqs(nfft, sample_rate), out64.cpu().numpy().astype(np.float64)


@dataclass
class WaterfallResult:
    """reference spectrum.py:48-52"""

    freqs: np.ndarray
    times: np.ndarray
    matrix: np.ndarray


class _WaterfallAggregator:
    """Bounded-memory accumulator of waterfall slices (reference spectrum.py:174-208): float32 dB rows are appended
    and, whenever there are more than ``max_slices``, averaged pairwise (in float64, the first time stamp of a pair
    kept).  The rows live in one device matrix; the pairwise pass is ``iqa_pair_average_rows``."""

    def __init__(self, *, max_slices: int, n_cols: int):
        self.max_slices = max(1, int(max_slices))
        self.n_cols = n_cols
        self._cap = self.max_slices + 1
        self._rows = D.empty(self._cap * n_cols, "float32")
        self._spare = D.empty(((self._cap + 1) // 2) * n_cols, "float32")
        self._n = 0
        self._times: list[float] = []

    def room(self) -> int:
        """Frames that may be appended before the next reduction has to run."""
        return self._cap - self._n

    def add_block(self, rows_f32_dev, times: list[float]) -> None:
        k = len(times)
        if k == 0:
            return
        assert k <= self.room()
        self._rows[self._n * self.n_cols : (self._n + k) * self.n_cols] = rows_f32_dev[: k * self.n_cols]
        self._n += k
        self._times.extend(float(t) for t in times)
        self._maybe_reduce()

    def _maybe_reduce(self) -> None:
        while self._n > self.max_slices:
            N.call("iqa_pair_average_rows", N.ptr(self._rows), c_int32(self._n), c_int32(self.n_cols), N.ptr(self._spare),
                   N.stream_ptr())
            half = (self._n + 1) // 2
            self._rows[: half * self.n_cols] = self._spare[: half * self.n_cols]
            self._times = self._times[0::2]
            self._n = half

    def finalize(self) -> tuple[np.ndarray, np.ndarray]:
        if self._n == 0:
            return np.empty(0, dtype=np.float32), np.empty((0, 0), dtype=np.float32)
        matrix = self._rows[: self._n * self.n_cols].cpu().numpy().reshape(self._n, self.n_cols).astype(np.float32, copy=False)
        return np.asarray(self._times, dtype=np.float32), matrix


def streaming_waterfall(chunks: Iterable, sample_rate: float, *, nfft: int, hop: int | None = None, max_slices: int = 400,
                        fft_workers: int | None = None):
    """Averaged PSD and waterfall slices from a stream of blocks -- reference spectrum.py:58-93 with the window
    iterator of :96-130: windows of ``nfft`` samples every ``hop`` (default nfft/4) across block boundaries, no
    trailing partial window.  Returns ``(freqs, avg_psd, WaterfallResult, frames)``; ``avg_psd`` is the mean of the
    frames' dB values (as the reference accumulates them)."""
    hop = max(1, hop or nfft // 4)
    eng = _PsdEngine(sample_rate=sample_rate, nfft=nfft)
    agg = _WaterfallAggregator(max_slices=max_slices, n_cols=nfft)
    psd_sum = D.zeros(nfft, "float64")
    frames = 0
    torch = D.torch_mod()
    pending = None  # device complex64: the tail of the stream that has not filled a window yet
    offset = 0      # global sample index of block[0] (spectrum.py:103-126)
    for chunk in chunks:
        if chunk is None:
            continue
        block = _as_complex_dev(chunk).reshape(-1)
        if block.numel() == 0:
            continue
        if pending is not None and pending.numel():
            block = torch.cat((pending, block))
            offset -= int(pending.numel())
        total = int(block.numel())
        if total < nfft:
            pending = block
            offset += total
            continue
        n_win = (total - nfft) // hop + 1
        done = 0
        while done < n_win:
            k = min(BATCH_FRAMES, n_win - done, agg.room())
            first = done * hop
            _, rows32 = eng.frames(block, total, first, hop, k, want_f64=False, want_f32=True, sum_db=psd_sum)
            agg.add_block(rows32, [(offset + first + i * hop) / sample_rate for i in range(k)])
            done += k
            frames += k
            if frames % 200 < k:
                LOG.debug("Accumulated %d FFT frames for waterfall preview.", frames)
        start = n_win * hop
        pending = block[start:].clone()
        offset += total - int(pending.numel())
    if frames == 0:
        raise ValueError("Input did not contain enough samples for one FFT frame.")
    avg_psd = (psd_sum.cpu().numpy() / frames).astype(np.float64)
    times, matrix = agg.finalize()
    freqs = _freqs(nfft, sample_rate)
    return freqs.copy(), avg_psd, WaterfallResult(freqs=freqs.copy(), times=times, matrix=matrix), frames
