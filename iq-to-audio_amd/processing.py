"""GPU stand-in for the reference's ``src/iq_to_audio/processing.py`` DSP surface.

Same names, arguments, state attributes and error behaviour as the reference classes
(cited per class), with all arithmetic in the HIP library (``libiqa_hotpath.so``):

    ProcessingConfig / ProcessingPipeline / ProcessingResult / ProcessingCancelled
    ComplexOscillator, OverlapSaveFIR, Decimator, design_channel_filter,
    choose_mix_sign, tune_chunk_size

plus :class:`Channelizer`, the fused ingest+mix+filter+decimate stage the pipeline
actually runs (one pass over the raw int16/u8/f32 capture in HBM).

Stage methods accept either NumPy arrays (NumPy comes back, like the reference) or
device tensors (device tensors come back -- no host round trip).  There is no CPU path:
without a GPU or without the built library every stage raises ``RuntimeError``.
"""
from __future__ import annotations

import contextlib
import logging
import math
import threading
from collections import OrderedDict
from ctypes import byref, c_double, c_int32, c_int64, c_void_p
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from . import _dev as D
from . import _native as N
from . import dsp_plan as P
from . import iqio
from .decoders import create_decoder
from .dsp_plan import design_channel_filter, tune_chunk_size  # noqa: F401  (re-exported API)
from .progress import PhaseState, ProgressSink, ProgressTracker

LOG = logging.getLogger(__name__)


@dataclass
class ProcessingConfig:
    """Identical field set and defaults to the reference (processing.py:38-62).
    ``fft_workers`` is accepted and ignored (there is no FFT on this path)."""

    in_path: Path
    target_freq: float = 0.0
    bandwidth: float = 12_500.0
    center_freq: float | None = None
    center_freq_source: str | None = None
    demod_mode: str = "nfm"
    fs_ch_target: float = 96_000.0
    deemph_us: float = 300.0
    agc_enabled: bool = True
    output_path: Path | None = None
    dump_iq_path: Path | None = None
    chunk_size: int = 1_048_576
    filter_block: int = 65_536
    iq_order: str = "iq"
    probe_only: bool = False
    mix_sign_override: int | None = None
    plot_stages_path: Path | None = None
    fft_workers: int | None = None
    max_input_seconds: float | None = None
    input_container: str | None = None
    input_format: str | None = None
    input_format_source: str | None = None
    input_sample_rate: float | None = None


@dataclass
class SampleRateProbe:
    """Where the sample rate came from (reference probe.py SampleRateProbe; only the header
    parse exists here -- no ffprobe / libsndfile)."""

    ffprobe: float | None = None
    header: float | None = None
    wave: float | None = None

    @property
    def value(self) -> float:
        for v in (self.ffprobe, self.header, self.wave):
            if v:
                return float(v)
        raise RuntimeError("Unable to determine sample rate.")


def _size(x) -> int:
    return int(x.numel()) if D.is_tensor(x) else int(np.asarray(x).size)


def _as_frames(raw, fmt: str):
    """(device tensor, n_frames) for raw capture frames: int16/uint8 interleaved pairs, or
    float32 pairs / complex64 for 'f32' (both are the same bytes)."""
    if fmt == "f32":
        is_c = raw.is_complex() if D.is_tensor(raw) else np.iscomplexobj(raw)
        if is_c:
            x = D.to_device(raw, "complex64")
            return x, int(x.numel())
        x = D.to_device(raw, "float32").reshape(-1)
        return x, int(x.numel()) // 2
    x = D.to_device(raw, {"s16": "int16", "u8": "uint8"}[fmt]).reshape(-1)
    return x, int(x.numel()) // 2


# --------------------------------------------------------------------------------------------- #
# pluggable stages                                                                              #
# --------------------------------------------------------------------------------------------- #


class ComplexOscillator:
    """Continuous complex exponential for frequency translation (reference processing.py:282-297).

    ``phase`` (radians) and ``increment`` are host floats exactly as in the reference; the
    float64 ramp ``phase + sign*increment*n`` is evaluated per sample on the GPU.
    ``fmt``/``iq_order`` let the stage also do the ingest convert when fed raw frames.
    """

    def __init__(self, freq_offset_hz: float, sample_rate: float):
        self.phase = 0.0
        self.increment = -2.0 * np.pi * freq_offset_hz / sample_rate

    def mix(self, samples, sign: int, *, fmt: str = "f32", iq_order: str = "iq"):
        if _size(samples) == 0:
            return samples
        if iq_order not in N.ORDER:
            raise ValueError(f"Unsupported iq_order '{iq_order}'")
        if fmt == "f32":
            x = D.to_device(samples, "complex64")
            n = x.numel()
        else:
            x = D.to_device(samples, {"s16": "int16", "u8": "uint8"}[fmt]).reshape(-1)
            n = x.numel() // 2
        out = D.empty(n, "complex64")
        step = sign * self.increment
        N.call("iqa_oscillator_mix", c_int32(P.FMT_CODE[fmt]), c_int32(N.ORDER[iq_order]), N.ptr(x), c_int64(n),
               c_double(self.phase), c_double(step), N.ptr(out), N.stream_ptr())
        self.phase = (self.phase + step * n) % (2.0 * np.pi)
        return D.like_input(out, samples)


class _ChannelKernel:
    """Shared launcher for the fused channelizer kernels.

    ``iqa_channelize`` (float32 VALU form, every format, guarded edges) is always available;
    for int16 captures with ceil(L/D) <= 64 the interior of each block runs on the int8-MFMA
    form ``iqa_channelize_mfma`` and only the few outputs that touch the history (head) or the
    end of the block (tail) go through the VALU kernel.
    """

    #: set to False to force the float32 VALU kernel everywhere (tests compare the two)
    use_mfma = True
    #: data path of the MFMA kernel: "ring" (channelize_ring.hip: persistent blocks stream their contiguous run of the
    #: capture through an LDS-DMA ring, tap fragments in registers; falls back to "plain" where it does not apply:
    #: D % 4 != 0, D > 256, multi-range passes) or "plain" (channelize_mfma.hip: per-lane row loads into VGPRs)
    mfma_variant = "ring"
    #: sums of the ring kernel: True (default) = one int32 256*S1 + S2 per output component with the tap unit enlarged
    #: until that cannot overflow for any input (~14-bit taps, error ~1e-5 of full scale); False = one int64
    #: (S1 << 32) + S2 with 16-bit taps -- the same integers as the per-lane kernel (~1e-6) -- at +10 % kernel time
    #: (ds_add_u64 moves 3 dwords and takes two passes through the LDS banks).  Both are exact integer sums.
    ring_acc32 = True
    RING_ROWS_KSTEPS = 11  # k steps per pass of the row-staged ring kernel (its tap fragments live in registers)
    _VARIANT = {"plain": (0, 0), "ring": (64, 0)}  # flags, extra LDS bytes
    mfma_min_outputs = 32768

    #: Precisions of a channelizer, cheapest first:
    #:   "fast"    -- the ring kernels, ONE int32 sum per output component, ~14-bit taps: z error ~3e-6 of full scale
    #:   "fine"    -- the same kernels, every tap-row group as TWO lanes (high-byte-only taps + their residue, added by
    #:                iqa_mfma_combine): twice the matrix work, error 10..90x smaller; uint8 captures: exact products
    #:   "full"    -- 16-bit taps without the int32 bound, the same two groups: z error ~1e-9 of full scale -- below the float32
    #:                rounding of z itself -- at ~2.3x the time of "fast".  Contiguous ring slots (D % 4 == 0, <= 15 k steps):
    #:                lanes of the ring kernels with 64-bit sums (shared ingest, pairs at 9..14 k steps); every other
    #:                decimation: chained passes of the per-lane kernel (separate S1/S2 sums).  int16 captures; uint8 ->
    #:                "fine", float32 -> "float32"
    #:   "float32" -- the float32 VALU kernel for every output (~20x the time of "fast"; the only form for float32 captures)
    PRECISIONS = ("fast", "fine", "full", "float32")

    def __init__(self, plan: P.ChannelPlan, exact: bool = False, precision: str | None = None):
        self.plan = plan
        precision = precision or ("float32" if exact else "fast")
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}, not {precision!r}")
        if plan.fmt == "f32" and precision != "fast":
            precision = "float32"
        if plan.fmt == "u8" and precision == "full":
            precision = "fine"
        self.precision = precision
        self.exact = precision == "float32"  # float32 kernel everywhere
        self.variant = self.mfma_variant  # ("full" off the contiguous ring slots: no ring mode below -> the per-lane kernel)
        self.acc32 = bool(self.ring_acc32) and precision != "full"
        self.residual = precision in ("fine", "full")
        lpad = int(N.lib().iqa_taps_padded_len(plan.ntaps))
        if plan.taps_window.size != lpad:
            raise ValueError("tap window padding does not match the library")
        self.taps_dev = D.from_numpy(plan.taps_window)
        self.params = N.ChanParams(
            fmt=P.FMT_CODE[plan.fmt], ntaps=plan.ntaps, decimation=plan.decimation, conj_sum=plan.conj_sum,
            rotate=plan.rotate, reserved=0, rot_step=plan.rot_step, rot_base=plan.rot_base,
            out_scale_re=float(np.real(plan.out_scale)), out_scale_im=float(np.imag(plan.out_scale)),
        )
        self.mfma = None  # planned lazily, the first time a block is long enough to use it
        self._lock = threading.RLock()  # kernels are shared through _KERNEL_CACHE; a launch mutates the pass parameters
        self.last_kernel = "k_channelize_v1"
        self._ring_bytes = int(N.lib().iqa_mfma_ring_bytes(plan.decimation)) if plan.fmt == "s16" else 0
        # which ring kernel covers this decimation: 1 = contiguous slots (all k steps in one pass), 2 = row-staged slots
        # (any D, k-step ranges of <= RING_ROWS_KSTEPS, int32 sums only; the only form for uint8 captures), 0 = none ->
        # the per-lane kernel (int16) or the VALU kernel (uint8)
        ks_all = -(-2 * plan.decimation // 32)
        self._ring_mode = 0
        if plan.fmt in ("s16", "u8") and self.variant == "ring":
            acc32, code = int(bool(self.acc32)), P.FMT_CODE[plan.fmt]
            self._ring_mode = int(N.lib().iqa_mfma_ring_mode(code, plan.decimation, 0, ks_all, acc32))
            if self._ring_mode == 0:
                self._ring_mode = int(N.lib().iqa_mfma_ring_mode(code, plan.decimation, 0, min(ks_all, self.RING_ROWS_KSTEPS), acc32))
        self._mfma_ok = bool(self.use_mfma and not self.exact and P.mfma_supported(plan) and (plan.fmt == "s16" or self._ring_mode == 2))

    def _ensure_mfma(self):
        with self._lock:
            return self._ensure_mfma_locked()

    def _ensure_mfma_locked(self):
        if self.mfma is None:
            ring = self.variant == "ring" and self._ring_mode != 0
            mp = P.plan_mfma(self.plan, acc32=ring and self.acc32,
                             max_ksteps=self.RING_ROWS_KSTEPS if self._ring_mode == 2 else None, residual=self.residual)
            self.mfma = mp
            self.afrag_dev = [D.from_numpy(g.afrag.reshape(-1).view(np.uint8)) for g in mp.groups]
            self.mfma_params = []
            for ps in mp.passes:
                # one 8-wave block per CU owns all 160 KiB of LDS: this pass's tap fragments + 16 B per output
                variant = self.variant
                if variant == "ring" and not ring:
                    variant = "plain"
                if self._range_max(ps.k_count, variant) < 512:  # the staging ring does not fit LDS next to the taps
                    variant = "plain"
                rng = self._range_max(ps.k_count, variant)
                self._pass_variant = getattr(self, "_pass_variant", []) + [variant]
                self.mfma_params.append(N.MfmaParams(
                    outputs_per_block=rng, reserved=self._VARIANT[variant][0] | (128 if (variant == "ring" and self.acc32) else 0) | (256 if (variant == "ring" and mp.groups[ps.group].high_only) else 0), unit=mp.groups[ps.group].unit / (256.0 if self.plan.fmt == "u8" else 1.0), c_re=ps.c_re,
                    c_im=ps.c_im, debug_stamps=None, q_group=mp.groups[ps.group].q, k_first=ps.k_first, k_count=ps.k_count,
                    finalize=0, partial_in_dev=None, partial_out_dev=None))
        return self.mfma

    def fixed_point_error_norm(self) -> float:
        """z error (RMS) of the fixed-point kernels per unit RMS of a white wideband input at full scale = 1: the 2-norm
        of the tap quantisation error (0.0 when this channel never runs on the matrix cores)."""
        return float(self._ensure_mfma().err_norm) if self._mfma_ok else 0.0

    def fixed_point_error_rms(self, wideband_rms: float) -> float:
        """Expected z error (RMS, fraction of full scale) of this kernel for a capture of the given wideband RMS: tap
        rounding x wideband level plus the level-independent floor of the dropped (low tap byte) x (low data byte)
        products (0.0 when this channel never runs on the matrix cores)."""
        return float(self._ensure_mfma().z_error_rms(wideband_rms)) if self._mfma_ok else 0.0

    def _range_max(self, k_count: int, variant: str) -> int:
        if variant == "ring":  # tap fragments in registers, sums in a sliding window: a block is not bounded by LDS
            return 1 << 24
        lds = 160 * 1024 - k_count * P.MFMA_KSTEP_BYTES - self._VARIANT[variant][1]
        return int(min(6144, (lds // 16 - 160) // 32 * 32))

    #: workgroups of a capture-long launch: one per CU of the MI355X
    launch_blocks = 256

    @classmethod
    def _block_outputs(cls, n_out: int, rmax: int, per_cu: int = 1) -> int:
        """Outputs per block for a launch of ``n_out`` outputs: as large as LDS allows, but chosen so that the
        number of blocks is a multiple of the 256 CUs (one block per CU, no ragged last round).  The ring kernel
        has no LDS bound (``rmax`` huge): every CU gets ONE contiguous range of the launch -- ``per_cu`` of them where
        that many of its workgroups fit into a CU's LDS (short rows: <= 3 k steps)."""
        blocks = cls.launch_blocks * per_cu
        rounds = max(1, -(-n_out // (blocks * rmax)))
        per = -(-n_out // (blocks * rounds))
        return int(min(rmax, max(512, -(-per // 32) * 32)))

    def _workgroups_per_cu(self, ps, variant: str) -> int:
        if variant != "ring":
            return 1
        lds = int(N.lib().iqa_mfma_ring_lds_bytes(P.FMT_CODE[self.plan.fmt], self.plan.decimation, ps.k_first, ps.k_count,
                                                  1 if self.acc32 else 0))
        return 2 if 0 < lds <= 80 * 1024 else 1

    def _valu(self, raw_dev, n_frames, consumed, hist_dev, m_first, n_out, out_dev):
        if n_out > 0:
            N.call("iqa_channelize", byref(self.params), N.ptr(self.taps_dev), N.ptr(raw_dev), c_int64(n_frames),
                   c_int64(consumed), N.ptr(hist_dev), c_int64(m_first), c_int64(n_out), N.ptr(out_dev), N.stream_ptr())

    def run(self, raw_dev, n_frames: int, consumed: int, hist_dev, m_first: int, n_out: int, out_dev=None,
            events=None, halo=None, edge_stream=None):
        """``events``: optional (start, stop) torch.cuda.Event pair recorded around the dominant launch.
        ``halo``: optional (buffer, lead_frames) -- ``raw_dev`` is the slice ``buffer[lead : lead + n_frames]`` (in
        frames) of a larger device buffer whose ``lead`` frames in front hold the history of this block (zeros at the
        start of a capture) and whose frames behind may be read (their values are never used): the matrix-core
        kernels then cover the block's first and last outputs too and the two VALU edge launches disappear.
        ``edge_stream``: optional torch stream for the small VALU launches of the block's first and last outputs (they
        write their own part of ``out_dev``); the caller orders it against the producers of ``raw_dev`` and the
        consumers of ``out_dev``."""
        with self._lock:
            return self._run(raw_dev, n_frames, consumed, hist_dev, m_first, n_out, out_dev, events, halo, edge_stream)

    def _interior(self, consumed: int, n_frames: int, m_first: int, n_out: int) -> tuple[int, int]:
        """Outputs [m_a, m_b) the matrix-core kernels can produce from this block's frames alone."""
        d = self.plan.decimation
        ksteps = -(-2 * d // 32)
        n_groups = max(1, -(-(-(-self.plan.ntaps // d)) // P.MFMA_Q))
        m_a, m_b = P.mfma_interior(consumed, n_frames, m_first, n_out, d, ksteps, n_groups)
        if self.variant == "ring" and self._ring_mode == 1:
            # a contiguous ring tile is fetched as 2048*ksteps bytes from its first frame
            m_b = min(m_b, (n_frames + consumed - 512 * ksteps - 1) // d + 2)
        return (m_a, m_b) if m_b > m_a else (m_first, m_first)

    def _mfma_passes(self, raw_dev, n_frames: int, consumed: int, m_a: int, n_int: int, out_dev, min_block: int = 512):
        mp = self._ensure_mfma()
        self.last_kernel = ("k_channelize_mfma_u8" if self.plan.fmt == "u8" else "k_channelize_mfma_s16") + (
            "_ring" if self._pass_variant[-1] == "ring" else "")
        partial = D.empty(2 * n_int, "float64") if len(mp.passes) > 1 else None
        for i, (ps, prm) in enumerate(zip(mp.passes, self.mfma_params)):
            last = i == len(mp.passes) - 1
            rng = self._block_outputs(n_int, self._range_max(ps.k_count, self._pass_variant[i]),
                                      self._workgroups_per_cu(ps, self._pass_variant[i]))
            if self._pass_variant[i] == "ring" and min_block < 512:  # short launches: more, smaller blocks
                rng = max(min_block, min(rng, -(-(-(-n_int // 256)) // 32) * 32))
            prm.outputs_per_block = rng
            prm.finalize = int(last)
            prm.partial_in_dev = partial.data_ptr() if (partial is not None and i > 0) else None
            prm.partial_out_dev = partial.data_ptr() if (partial is not None and not last) else None
            afrag = self.afrag_dev[ps.group][ps.k_first * P.MFMA_KSTEP_BYTES :]
            N.call("iqa_channelize_mfma", byref(self.params), byref(prm), N.ptr(afrag), N.ptr(raw_dev),
                   c_int64(n_frames), c_int64(consumed), c_int64(m_a), c_int64(n_int), N.ptr(out_dev), N.stream_ptr())

    def run_interior_only(self, raw_dev, n_frames: int, m_first: int, n_out: int, out_dev) -> bool:
        """Outputs [m_first, m_first + n_out) of a block that starts the capture (consumed = 0), matrix-core kernels
        only -- for callers that do not want the outputs near the block's edges (the mixer-sign probes discard the
        filter's transient and read a snippet of a longer buffer).  False (nothing launched) when the range is not
        wholly interior or the capture format has no matrix-core kernel."""
        with self._lock:
            if not (self._mfma_ok and self.variant == "ring" and self._ring_mode) or n_out < 64:
                return False
            m_a, m_b = self._interior(0, n_frames, m_first, n_out)
            if m_a != m_first or m_b != m_first + n_out:
                return False
            self._mfma_passes(raw_dev, n_frames, 0, m_first, n_out, out_dev, min_block=64)
            return True

    def _edges(self, edge_stream, *args):
        if edge_stream is None:
            return self._valu(*args)
        with D.torch_mod().cuda.stream(edge_stream):
            self._valu(*args)

    def _run(self, raw_dev, n_frames: int, consumed: int, hist_dev, m_first: int, n_out: int, out_dev, events, halo=None,
             edge_stream=None):
        if out_dev is None:
            out_dev = D.empty(n_out, "complex64")
        self.last_kernel = "k_channelize_v1"
        if self._mfma_ok and n_out >= self.mfma_min_outputs:
            big, big_frames, big_consumed = raw_dev, n_frames, consumed
            if halo is not None:  # the matrix-core kernels address the enclosing buffer
                big, lead = halo
                big_frames, big_consumed = int(big.numel()) // 2, consumed - int(lead)  # 2 values per frame (I, Q)
            m_a, m_b = self._interior(big_consumed, big_frames, m_first, n_out)
            if m_b - m_a >= self.mfma_min_outputs:
                self._edges(edge_stream, raw_dev, n_frames, consumed, hist_dev, m_first, m_a - m_first, out_dev)
                if events:
                    events[0].record()
                self._mfma_passes(big, big_frames, big_consumed, m_a, m_b - m_a, out_dev[m_a - m_first :])
                if events:
                    events[1].record()
                self._edges(edge_stream, raw_dev, n_frames, consumed, hist_dev, m_b, m_first + n_out - m_b,
                            out_dev[m_b - m_first :])
                return out_dev
        if events:
            events[0].record()
        self._valu(raw_dev, n_frames, consumed, hist_dev, m_first, n_out, out_dev)
        if events:
            events[1].record()
        return out_dev


# Planned channelizer kernels (rotated taps, quantised MFMA fragments, their device copies) are immutable once
# built and cost ~0.5 ms of host NumPy per configuration: a batch of captures with the same settings, the two
# probes of choose_mix_sign and the channelizer that follows them all share them through this small LRU.
_KERNEL_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()
_KERNEL_CACHE_MAX = 192  # (BASELINE config 5 on one GPU: 40 channels x (two probe signs + the channel) = 120 kernels, ~0.5 MB of device taps each)
_KERNEL_CACHE_LOCK = threading.Lock()


_TAPS_MEMO: dict = {}  # id(array) -> (array, its bytes, their hash), for arrays that cannot change


def _taps_fingerprint(taps: np.ndarray):
    """(bytes, hash) of a tap vector.  Hashing 50-260 KB costs 25-130 us, and a batch asks three times per capture with
    the same array: an array that owns its data and is not writeable (``immutable_taps``) is fingerprinted once."""
    frozen = (not taps.flags.writeable) and taps.base is None
    if frozen:
        memo = _TAPS_MEMO.get(id(taps))
        if memo is not None and memo[0] is taps:
            return memo[1], memo[2]
    raw = taps.tobytes()
    h = hash(raw)
    if frozen:
        with _KERNEL_CACHE_LOCK:
            if len(_TAPS_MEMO) >= 64:
                _TAPS_MEMO.clear()
            _TAPS_MEMO[id(taps)] = (taps, raw, h)
    return raw, h


def immutable_taps(taps) -> np.ndarray:
    """A private, contiguous, read-only copy of a tap vector (what the kernel cache can recognise without hashing)."""
    out = np.array(taps, copy=True, order="C")
    out.setflags(write=False)
    return out


def _cached_kernel(taps: np.ndarray, *, sample_rate: float, freq_offset: float, mix_sign: int, decimation: int,
                   fmt: str, iq_order: str, exact: bool = False, precision: str | None = None):
    """(plan, kernel) for this configuration, planned once per process and device."""
    taps = np.ascontiguousarray(taps)
    raw, raw_hash = _taps_fingerprint(taps)
    key = (raw_hash, taps.dtype.str, taps.shape, float(sample_rate), float(freq_offset), int(mix_sign), int(decimation),
           fmt, iq_order, _ChannelKernel.use_mfma, _ChannelKernel.mfma_variant, _ChannelKernel.ring_acc32,
           precision or ("float32" if exact else "fast"), D.torch_mod().cuda.current_device())
    with _KERNEL_CACHE_LOCK:
        hit = _KERNEL_CACHE.get(key)
        if hit is not None and (hit[0] is raw or hit[0] == raw):
            _KERNEL_CACHE.move_to_end(key)
            return hit[1], hit[2]
    lpad = int(N.lib().iqa_taps_padded_len(len(taps)))
    plan = P.plan_channel(taps, sample_rate=sample_rate, freq_offset=freq_offset, mix_sign=mix_sign,
                          decimation=decimation, fmt=fmt, iq_order=iq_order, padded_len=lpad)
    kernel = _ChannelKernel(plan, exact, precision)
    with _KERNEL_CACHE_LOCK:
        _KERNEL_CACHE[key] = (raw, plan, kernel)
        while len(_KERNEL_CACHE) > _KERNEL_CACHE_MAX:
            _KERNEL_CACHE.popitem(last=False)
    return plan, kernel


class OverlapSaveFIR:
    """Streaming channel filter stage (reference processing.py:300-346).

    Same constructor, attributes (``taps``, ``filter_len``, ``overlap``, ``block_size``,
    ``fft_size``, ``state``) and semantics -- causal linear convolution, zero initial state,
    output length == input length, history of the last L-1 input samples carried across
    calls -- but evaluated as a direct time-domain dot product on the GPU: ``block_size``
    and ``fft_size`` are kept for API compatibility and do not affect the result.
    """

    def __init__(self, taps: np.ndarray, block_size: int, *, workers: int | None = None):
        if block_size <= 0:
            raise ValueError("block_size must be positive")
        self.taps = np.asarray(taps).astype(np.complex128)
        self.filter_len = len(taps)
        self.overlap = self.filter_len - 1
        self.block_size = block_size
        self.fft_size = 1 << math.ceil(math.log2(self.block_size + self.filter_len - 1))
        self.workers = None
        self._kernel = None
        self._real_taps = np.asarray(taps, dtype=np.float64)
        self._hist = None  # device complex64[L-1]
        self._consumed = 0

    @property
    def state(self) -> np.ndarray:
        if self._hist is None:
            return np.zeros(self.overlap, dtype=np.complex64)
        return self._hist.cpu().numpy()

    @property
    def taps_fft(self) -> np.ndarray:
        """The reference's frequency response of the zero-padded taps (processing.py:317-321), computed on request:
        nothing here uses it (the filter runs in the time domain), it exists for callers that inspect the stage."""
        padded = np.zeros(self.fft_size, dtype=np.complex128)
        padded[: self.filter_len] = self.taps
        return np.fft.fft(padded)

    def process(self, samples):
        if _size(samples) == 0:
            return samples
        if self._kernel is None:
            self._kernel = _ChannelKernel(P.plan_plain_fir(self._real_taps))
        x = D.to_device(samples, "complex64")
        n = x.numel()
        y = self._kernel.run(x, n, self._consumed, self._hist, self._consumed, n)
        if self.overlap:
            nxt = D.empty(self.overlap, "complex64")
            N.call("iqa_history_update", c_int32(P.FMT_CODE["f32"]), c_int32(self.filter_len), N.ptr(self._hist),
                   N.ptr(x), c_int64(n), N.ptr(nxt), N.stream_ptr())
            self._hist = nxt
        self._consumed += n
        return D.like_input(y, samples)


class Decimator:
    """Keep global sample indices 0, D, 2D, ... across calls (reference processing.py:349-360)."""

    def __init__(self, factor: int):
        self.factor = max(1, factor)
        self.offset = 0

    def process(self, samples):
        n = _size(samples)
        if self.factor == 1 or n == 0:
            return samples
        start = (-self.offset) % self.factor
        self.offset = (self.offset + n) % self.factor
        n_out = 0 if start >= n else -(-(n - start) // self.factor)
        x = D.to_device(samples, "complex64")
        out = D.empty(n_out, "complex64")
        N.call("iqa_decimate", N.ptr(x), c_int64(n), c_int64(start), c_int32(self.factor), N.ptr(out), c_int64(n_out),
               N.stream_ptr())
        return D.like_input(out, samples)


class Channelizer:
    """Fused ingest + NCO mix + channel FIR + decimate over raw capture frames.

    Equivalent to ``Decimator(D).process(OverlapSaveFIR(taps, B).process(
    ComplexOscillator(f_off, fs).mix(ingest(raw), sign)))`` of the reference
    (processing.py:1088-1096) with the streaming state of all three carried across calls,
    computed in one kernel that reads each raw frame (4 bytes for int16 I/Q) from HBM and
    writes only the decimated complex64 stream.
    """

    def __init__(self, taps: np.ndarray, *, sample_rate: float, freq_offset: float, mix_sign: int, decimation: int,
                 fmt: str = "s16", iq_order: str = "iq", exact: bool = False, precision: str | None = None):
        """``precision``: "fast" (default), "fine", "full" or "float32" -- see ``_ChannelKernel.PRECISIONS``; what the
        pipeline's precision guard and its SSB-with-AGC rule pick per target.  ``exact=True`` is "float32"."""
        self.plan, self._kernel = _cached_kernel(taps, sample_rate=sample_rate, freq_offset=freq_offset, mix_sign=mix_sign,
                                                 decimation=decimation, fmt=fmt, iq_order=iq_order, exact=exact, precision=precision)
        self.precision = self._kernel.precision
        self.fmt = fmt
        self.decimation = int(decimation)
        self.ntaps = len(taps)
        self.consumed = 0  # frames seen so far (global index of the next frame)
        self._hist = None  # device raw frames [L-1], same fmt

    def plan_ahead(self) -> None:
        """Do the host-side MFMA planning and tap upload now (otherwise done lazily by the first long block)."""
        if self._kernel._mfma_ok:
            self._kernel._ensure_mfma()

    def outputs_for(self, n_frames: int) -> tuple[int, int]:
        """(m_first, n_out) for a block of ``n_frames`` frames appended now."""
        d = self.decimation
        m_first = -(-self.consumed // d)
        m_end = -(-(self.consumed + n_frames) // d)
        return m_first, m_end - m_first

    def process(self, raw, out_dev=None, events=None, last_block: bool = False, halo=None, edge_stream=None):
        """``raw``: interleaved frames (NumPy or device tensor, dtype of ``fmt``; complex64 for f32).
        Returns the decimated complex64 samples for this block.  ``last_block``: nothing follows, so the
        L-1 frame history is not carried over (saves a launch for whole-capture calls).  ``halo``, ``edge_stream``:
        see ``_ChannelKernel.run`` (int16 / uint8 device captures only)."""
        x, n = _as_frames(raw, self.fmt)
        if n == 0:
            return D.like_input(D.empty(0, "complex64"), raw)
        m_first, n_out = self.outputs_for(n)
        if halo is not None and (self.fmt not in ("s16", "u8") or not D.is_tensor(raw)):
            halo = None
        z = None
        if n_out and D.is_tensor(raw) and self._several_lanes():
            # a filter with several tap-row groups: its groups as lanes of ONE shared-ingest launch (in pairs where the
            # kernel offers them) + the combine launch, instead of one pass over the capture per group
            if events:
                events[0].record()
            zs = ChannelBank([self])._run_shared(x, n, m_first, n_out, [out_dev], halo, edge_stream)
            if zs is not None:
                z = zs[0]
                if events:
                    events[1].record()
        if z is None:
            z = (self._kernel.run(x, n, self.consumed, self._hist, m_first, n_out, out_dev, events, halo, edge_stream)
                 if n_out else D.empty(0, "complex64"))
        self._advance(x, n, last_block)
        return D.like_input(z, raw)

    def _several_lanes(self) -> bool:
        k = self._kernel
        if not (self.lanes_for_groups and self.fmt in ("s16", "u8") and ChannelBank._lane_capable(k)):
            return False
        return len(k._ensure_mfma().groups) > 1

    lanes_for_groups = True  # (class switch: profiles compare against the chained passes)

    def _advance(self, x, n: int, last_block: bool = False) -> None:
        """Carry the last L-1 raw frames over to the next block and move on by ``n`` frames."""
        keep = 0 if last_block else self.ntaps - 1
        if keep:
            nxt = D.empty(keep * iqio.FRAME_BYTES[self.fmt], "uint8")
            N.call("iqa_history_update", c_int32(P.FMT_CODE[self.fmt]), c_int32(self.ntaps), N.ptr(self._hist),
                   N.ptr(x), c_int64(n), N.ptr(nxt), N.stream_ptr())
            self._hist = nxt
        self.consumed += n


class ChannelBank:
    """Several channels of ONE capture through a single pass over each block (BASELINE configs 3 and 5).

    The reference runs one whole pipeline per ``--ft`` target over the same file (cli.py:683-710).  Here the channelizers
    of a capture that share the decimation and the sample format put all their (channel, tap-row group) pairs into ONE
    launch of the ring kernel (``iqa_channelize_mfma_multi``): the lanes of a stretch of the capture run at the same time
    on the CUs of one XCD, so the stretch is fetched from HBM once and the other lanes read it from that XCD's L2.
    Filters with several tap-row groups (ceil(L/D) > 64) are several lanes whose partial sums ``iqa_mfma_combine`` adds
    up in group order -- the same additions, in the same order, as the chained single-channel passes, so a bank
    produces exactly what its channelizers would produce one by one.  The few outputs at a block's head and tail go
    through each channel's float32 kernel as usual.  Falls back to one channel at a time whenever a block is too short
    for the matrix-core kernels or the channels do not share a kernel shape.
    """

    MAX_LANES = 16  # per launch (the lane table travels as kernel arguments)
    skip_zero_low_taps = bool(int(__import__("os").environ.get("IQA_SKIP_ZERO_LOW_TAPS", "1")))  # (A/B switch for profiles/)
    pair_lanes = True  # two lanes of equal tap-row group per workgroup where the kernel offers it (see _run_shared)

    def __init__(self, channelizers: list):
        if not channelizers:
            raise ValueError("a bank needs at least one channelizer")
        self.chans = list(channelizers)
        first = self.chans[0]
        self.fmt, self.decimation = first.fmt, first.decimation
        for c in self.chans:
            if (c.fmt, c.decimation, c.consumed) != (self.fmt, self.decimation, first.consumed):
                raise ValueError("the channels of a bank share the capture: same sample format, decimation and position")
        self.last_launch = None  # {"lanes": n, "launches": n, "combines": n} of the most recent block (None: one by one)

    @staticmethod
    def _lane_capable(k) -> bool:
        """This channel's tap-row groups can be lanes of a shared-ingest launch: ring kernels with int32 sums (any slot
        form) or with 64-bit sums (contiguous slots only)."""
        return bool(k._mfma_ok and k.variant == "ring" and k._ring_mode and (k.acc32 or k._ring_mode == 1))

    def _shared_shape(self) -> bool:
        ks = [c._kernel for c in self.chans]
        if not ks or not all(self._lane_capable(k) for k in ks) or len({k.acc32 for k in ks}) != 1:
            return False
        if len(ks) == 1:  # one channel: worth a shared-ingest launch only when its filter is several lanes (tap-row groups)
            return len(ks[0]._ensure_mfma().groups) > 1
        return len({k._ring_mode for k in ks}) == 1

    def process(self, raw, outs=None, last_block: bool = False, halo=None, edge_stream=None) -> list:
        """One block of the capture for every channel; returns the decimated streams in channel order.
        ``edge_stream``: optional torch stream for the small float32 launches of every channel's first and last outputs
        (they write their own part of the outputs); the caller orders it against the producers of ``raw`` and the
        consumers of the outputs, as with ``Channelizer.process``."""
        x, n = _as_frames(raw, self.fmt)
        outs = list(outs) if outs is not None else [None] * len(self.chans)
        first = self.chans[0]
        m_first, n_out = first.outputs_for(n)
        zs = None
        if n and n_out and D.is_tensor(raw) and self._shared_shape():
            zs = self._run_shared(x, n, m_first, n_out, outs, halo, edge_stream)
        if zs is None:
            self.last_launch = None
            capable = [i for i, c in enumerate(self.chans) if self._lane_capable(c._kernel)]
            widths = sorted({self.chans[i]._kernel.acc32 for i in capable}, reverse=True)
            if D.is_tensor(raw) and capable and (len(capable) < len(self.chans) or len(widths) > 1):
                # channels of several precisions in the bank: the lanes with int32 sums share one pass, the lanes with
                # 64-bit sums ("full") another, whatever has no lanes ("full" off the contiguous slots, "float32") follows
                # one by one
                res = [None] * len(self.chans)
                self.last_launch, self.launches = None, []
                for acc32 in widths:
                    lanes = [i for i in capable if self.chans[i]._kernel.acc32 == acc32]
                    sub = ChannelBank([self.chans[i] for i in lanes])
                    sub.combines_on_edge_stream = self.combines_on_edge_stream
                    got = sub.process(raw, outs=[outs[i] for i in lanes], last_block=last_block, halo=halo, edge_stream=edge_stream)
                    self.launches.append(sub.last_launch)
                    if self.last_launch is None:
                        self.last_launch = sub.last_launch
                    for i, z in zip(lanes, got):
                        res[i] = z
                for i, (c, o) in enumerate(zip(self.chans, outs)):
                    if res[i] is None:
                        res[i] = c.process(raw, out_dev=o, last_block=last_block, halo=halo)
                return res
            return [c.process(raw, out_dev=o, last_block=last_block, halo=halo) for c, o in zip(self.chans, outs)]
        for c in self.chans:
            c._advance(x, n, last_block)
        return zs

    def run_interior_only(self, x_all, n_frames: int, m_first: int, n_out: int, outs: list) -> bool:
        """Outputs [m_first, m_first + n_out) of a block that starts the capture, for every channel, matrix-core kernels
        only (no float32 edge launches): what ``_ChannelKernel.run_interior_only`` does for one channel, for callers that
        do not want the outputs near the block's edges (the mixer-sign probes).  False (nothing launched) when the range
        is not interior for every channel or the channels do not share a kernel shape."""
        if n_out < 64 or not self._shared_shape():
            return False
        kernels = [c._kernel for c in self.chans]
        plans = [k._ensure_mfma() for k in kernels]
        if len(kernels) > self.MAX_LANES or any(len(mp.groups) != 1 or len(mp.passes) != 1 for mp in plans):
            # filters with several tap-row groups / k-step passes: partial sums and combine launches as for a whole block
            return self._run_shared(x_all, n_frames, m_first, n_out, list(outs), None, None, interior_only=True) is not None
        # single-group, single-pass filters (the usual probe): one lane each, one launch, nothing else
        if any(k._interior(0, n_frames, m_first, n_out) != (m_first, m_first + n_out) for k in kernels):
            return False
        if len({(mp.passes[0].k_first, mp.passes[0].k_count) for mp in plans}) != 1:
            return False
        ps0 = plans[0].passes[0]
        ranges = 8 * max(1, (_ChannelKernel.launch_blocks // 8) // len(kernels))
        rng = max(64, -(-(-(-n_out // ranges)) // 32) * 32)
        table = (N.MfmaLane * len(kernels))()
        for lane, k, mp, z in zip(table, kernels, plans, outs):
            ps = mp.passes[0]
            lane.afrag_dev = k.afrag_dev[0][ps.k_first * P.MFMA_KSTEP_BYTES :].data_ptr()
            lane.z_out_dev = z.data_ptr()
            lane.partial_in_dev = lane.partial_out_dev = None
            lane.unit = mp.groups[0].unit / (256.0 if self.fmt == "u8" else 1.0)
            lane.c_re, lane.c_im = ps.c_re, ps.c_im
            lane.rot_step, lane.rot_base = k.params.rot_step, k.params.rot_base
            lane.out_scale_re, lane.out_scale_im = k.params.out_scale_re, k.params.out_scale_im
            lane.q_group, lane.finalize = mp.groups[0].q, 1
            lane.conj_sum, lane.rotate = k.params.conj_sum, k.params.rotate
            lane.raw_partials = 0
            k.last_kernel = ("k_channelize_mfma_u8" if self.fmt == "u8" else "k_channelize_mfma_s16") + "_ring"
        N.call("iqa_channelize_mfma_multi", c_int32(P.FMT_CODE[self.fmt]), c_int32(self.decimation), c_int32(ps0.k_first),
               c_int32(ps0.k_count), c_int32(rng), table, c_int32(len(kernels)), N.ptr(x_all), c_int64(n_frames), c_int64(0),
               c_int64(m_first), c_int64(n_out), N.stream_ptr())
        self.last_launch = dict(lanes=len(kernels), launches=1, combines=0, pairs=0)
        return True

    combines_on_edge_stream = True  # (False: the combine launches stay on the caller's stream, only the float32 edge launches go to ``edge_stream``)

    def _run_shared(self, x, n: int, m_first: int, n_out: int, outs: list, halo, edge_stream=None, interior_only: bool = False):
        kernels = [c._kernel for c in self.chans]
        consumed = self.chans[0].consumed
        big, big_frames, big_consumed = x, n, consumed
        if halo is not None:
            big, lead = halo
            big_frames, big_consumed = int(big.numel()) // 2, consumed - int(lead)
        spans = [k._interior(big_consumed, big_frames, m_first, n_out) for k in kernels]
        m_a, m_b = max(s[0] for s in spans), min(s[1] for s in spans)
        if interior_only:
            if any(s != (m_first, m_first + n_out) for s in spans):
                return None
        elif any(s[1] <= s[0] for s in spans) or m_b - m_a < _ChannelKernel.mfma_min_outputs:
            return None
        n_int = m_b - m_a
        plans = [k._ensure_mfma() for k in kernels]
        if len({tuple((ps.k_first, ps.k_count) for ps in mp.passes if ps.group == 0) for mp in plans}) != 1:
            return None
        zs = [o if o is not None else D.empty(n_out, "complex64") for o in outs]
        for c, k, z in zip(self.chans, kernels, zs):  # each channel's own edges (history in front, end of block behind)
            if not interior_only:
                k._edges(edge_stream, x, n, consumed, c._hist, m_first, m_a - m_first, z)
                k._edges(edge_stream, x, n, consumed, c._hist, m_b, m_first + n_out - m_b, z[m_b - m_first :])
            k.last_kernel = ("k_channelize_mfma_u8" if self.fmt == "u8" else "k_channelize_mfma_s16") + "_ring"
        kranges = [(ps.k_first, ps.k_count) for ps in plans[0].passes if ps.group == 0]
        ids = [(ci, gi) for ci, mp in enumerate(plans) for gi in range(len(mp.groups))]  # lane identities
        need_partial = {(ci, gi): (len(kranges) > 1 or len(plans[ci].groups) > 1) for ci, gi in ids}
        # single k-step range: a tap-row group's partial sums travel as the exact int32 pairs (8 B per output) and the
        # combine kernel scales them; chained k-step ranges -- and lanes with 64-bit sums -- hand on double2 sums (16 B)
        acc32 = bool(kernels[0].acc32)
        raw = len(kranges) == 1 and acc32
        partial = {key: D.empty(2 * n_int, "int32" if raw else "float64") for key, needed in need_partial.items() if needed}
        cpx = max(1, _ChannelKernel.launch_blocks // 8)  # CUs per XCD class the launch may fill
        launches = 0

        def launch(part, entry: str, units: int) -> None:
            """One launch per k-step range for the lanes ``part``; ``units`` workgroups share a stretch of the capture."""
            nonlocal launches
            ranges = 8 * max(1, cpx // units)
            rng = max(128, -(-(-(-n_int // ranges)) // 32) * 32)
            for ri, (k_first, k_count) in enumerate(kranges):
                last_range = ri == len(kranges) - 1
                table = (N.MfmaLane * len(part))()
                for lane, ident in zip(table, part):
                    if ident is None:  # the empty half of an odd pair: a zeroed entry
                        continue
                    ci, gi = ident
                    k, mp = kernels[ci], plans[ci]
                    ps = next(p_ for p_ in mp.passes if p_.group == gi and p_.k_first == k_first)
                    fin = last_range and len(mp.groups) == 1
                    buf = partial.get((ci, gi))
                    lane.afrag_dev = k.afrag_dev[gi][k_first * P.MFMA_KSTEP_BYTES :].data_ptr()
                    lane.z_out_dev = zs[ci][m_a - m_first :].data_ptr() if fin else None
                    lane.partial_in_dev = buf.data_ptr() if (buf is not None and ri > 0) else None
                    lane.partial_out_dev = None if fin else buf.data_ptr()
                    lane.unit = mp.groups[gi].unit / (256.0 if self.fmt == "u8" else 1.0)
                    lane.c_re, lane.c_im = ps.c_re, ps.c_im
                    lane.rot_step, lane.rot_base = k.params.rot_step, k.params.rot_base
                    lane.out_scale_re, lane.out_scale_im = k.params.out_scale_re, k.params.out_scale_im
                    lane.q_group, lane.finalize = mp.groups[gi].q, int(fin)
                    lane.conj_sum, lane.rotate = k.params.conj_sum, k.params.rotate
                    lane.raw_partials = int(raw and not fin)
                    lane.reserved = (0 if acc32 else 1) | (2 if (mp.groups[gi].high_only and self.skip_zero_low_taps) else 0)  # (bit 0: 64-bit sums; bit 1: q2 == 0)
                N.call(entry, c_int32(P.FMT_CODE[self.fmt]), c_int32(self.decimation), c_int32(k_first), c_int32(k_count), c_int32(rng),
                       table, c_int32(len(part)), N.ptr(big), c_int64(big_frames), c_int64(big_consumed), c_int64(m_a), c_int64(n_int),
                       N.stream_ptr())
                launches += 1

        # Two lanes to a workgroup where the kernel offers it (contiguous ring slots without loader waves: 9..16 k steps):
        # both read every staged tile of the capture -- half the L2 -> LDS traffic per lane and a ring twice as deep in
        # rounds.  A pair's first lane has the larger (or the same) tap-row group: lanes in descending group order, two
        # by two; an odd lane out shares its workgroup with nobody (None).  ONE launch either way: the capture crosses
        # HBM once.
        if self.pair_lanes and len(kranges) == 1 and (N.lib().iqa_mfma_ring_lanes(P.FMT_CODE[self.fmt], self.decimation, *kranges[0], int(acc32)) & 2):
            paired = sorted(ids, key=lambda i: -plans[i[0]].groups[i[1]].q)
            if len(paired) & 1:
                paired.append(None)
            for lo in range(0, len(paired), self.MAX_LANES):
                part = paired[lo : lo + self.MAX_LANES]
                launch(part, "iqa_channelize_mfma_pairs", len(part) // 2)
            n_pairs = len(paired) // 2
        else:
            n_pairs = 0
            for lo in range(0, len(ids), self.MAX_LANES):
                part = ids[lo : lo + self.MAX_LANES]
                launch(part, "iqa_channelize_mfma_multi", len(part))
        main = None
        if edge_stream is not None and self.combines_on_edge_stream and any(len(mp.groups) > 1 for mp in plans):
            # the combine launches go where the consumers of the outputs are queued (behind the pass): the caller's stream
            # then holds the pass alone.  (No stream calls at all otherwise: this function also runs inside graph captures,
            # where a set_stream -- even to the current stream -- made the replays 2.5x slower.)
            torch = D.torch_mod()
            main = torch.cuda.current_stream()
            passed = torch.cuda.Event()
            passed.record(main)
            edge_stream.wait_event(passed)
            for t in partial.values():
                t.record_stream(edge_stream)
            torch.cuda.set_stream(edge_stream)
        try:
            combines = self._combine(plans, kernels, partial, raw, m_a, m_first, n_int, zs)
        finally:
            if main is not None:
                D.torch_mod().cuda.set_stream(main)
        self.last_launch = dict(lanes=len(ids), launches=launches, combines=combines, pairs=n_pairs)
        return zs

    def _combine(self, plans, kernels, partial, raw, m_a, m_first, n_int, zs) -> int:
        combines = 0
        for ci, mp in enumerate(plans):
            if len(mp.groups) > 1:
                ptrs = (c_void_p * len(mp.groups))(*[partial[(ci, gi)].data_ptr() for gi in range(len(mp.groups))])
                scale = None
                if raw:
                    vals = []
                    for gi in range(len(mp.groups)):
                        ps = next(p_ for p_ in mp.passes if p_.group == gi)
                        vals += [mp.groups[gi].unit / (256.0 if self.fmt == "u8" else 1.0), ps.c_re, ps.c_im]
                    scale = (c_double * len(vals))(*vals)
                N.call("iqa_mfma_combine", byref(kernels[ci].params), ptrs, c_int32(len(mp.groups)), scale, c_int64(m_a),
                       c_int64(n_int), N.ptr(zs[ci][m_a - m_first :]), N.stream_ptr())
                combines += 1
        return combines


def probe_targets(warmup, sample_rate: float, specs: list, decimation: int, *, fmt: str, iq_order: str, host) -> list | None:
    """The mixer-sign probes of SEVERAL targets of one capture (``specs``: ``(freq_offset, taps)`` each) in as few launches
    as one bank takes: both signs of every target are channels of one :class:`ChannelBank` over the snippet (lane pairs,
    tap-row groups and their combine launches as for the capture itself) and ONE reduction writes all mean powers into
    ``host`` (pinned float64, two per target, owned by the caller until the probes have been read).  Returns one
    :class:`MixSignProbe` per target, or None when the grouped path does not apply (the caller then probes target by
    target): float32 captures, targets whose snippet or discard lengths differ, kernels without a shared shape."""
    if fmt not in ("s16", "u8") or not specs:
        return None
    x_all, n_in = _as_frames(warmup, fmt)
    decim = max(decimation, 1)
    shapes = set()
    for _, taps in specs:  # the lengths MixSignProbe.__init__ / _probe_one derive (reference processing.py:636-656)
        ntaps = len(taps)
        snippet = min(n_in, max(int(sample_rate * 0.05), ntaps * 4, 131_072))
        if snippet < ntaps:
            snippet = min(n_in, ntaps * 2)
        n_z = -(-snippet // decim)
        discard = min(ntaps, n_z // 4)
        shapes.add((n_z, discard if n_z - discard else 0))
    if len(shapes) != 1:
        return None
    n_z, discard = shapes.pop()
    keep = n_z - discard
    if keep < 64 or keep > MixSignProbe.DIRECT_MAX or host.numel() < 2 * len(specs):
        return None
    chans = [Channelizer(taps, sample_rate=sample_rate, freq_offset=f_off, mix_sign=sign, decimation=decim, fmt=fmt, iq_order=iq_order)
             for f_off, taps in specs for sign in (1, -1)]
    z_keep = D.empty(len(chans) * keep, "complex64")
    if not ChannelBank(chans).run_interior_only(x_all, n_in, discard, keep, [z_keep[i * keep : (i + 1) * keep] for i in range(len(chans))]):
        return None
    N.call("iqa_mean_power_batch", N.ptr(z_keep), c_int64(keep), c_int32(len(chans)), c_int64(0), N.ptr(host), N.stream_ptr())
    level_slot = None
    if host.numel() > 2 * len(specs):  # room for the wideband level of the warm-up block behind the powers
        level_slot = host[2 * len(specs) : 2 * len(specs) + 1]
        queue_raw_level(warmup, fmt, level_slot)
    done = D.torch_mod().cuda.Event()
    done.record()
    return [MixSignProbe.from_powers(host[2 * i : 2 * i + 2], done, level_slot, fmt) for i in range(len(specs))]


def _mean_power_into(z_dev, skip: int, out_slot) -> None:
    N.call("iqa_mean_power", N.ptr(z_dev), c_int64(z_dev.numel()), c_int64(skip), N.ptr(out_slot), N.stream_ptr())


_PINNED_SCALARS: list = []  # [pinned double[4], owner] -- a buffer goes back to the pool when its probe has been read


def _pinned_scalars(owner):
    """A reusable pinned double[4] for probe read-backs (pin_memory() is slow: allocate once per slot).  The
    buffer stays with ``owner`` until ``_release_scalars``: several probes may be in flight at once."""
    torch = D.torch_mod()
    with _KERNEL_CACHE_LOCK:
        for t in _PINNED_SCALARS:
            if t[1] is None:
                t[1] = owner
                return t[0]
        t = [torch.zeros(4, dtype=torch.float64).pin_memory(), owner]  # [power(+1), power(-1), raw mean square, spare]
        _PINNED_SCALARS.append(t)
        return t[0]


def reserve_pinned_scalars(count: int) -> None:
    """Make sure ``count`` probe read-back buffers are free NOW (pinning memory is not allowed while a stream is being
    captured: a caller about to capture ``count`` probes into a graph reserves them first)."""
    torch = D.torch_mod()
    with _KERNEL_CACHE_LOCK:
        free = sum(1 for t in _PINNED_SCALARS if t[1] is None)
        for _ in range(max(0, count - free)):
            _PINNED_SCALARS.append([torch.zeros(4, dtype=torch.float64).pin_memory(), None])


def _release_scalars(buf) -> None:
    with _KERNEL_CACHE_LOCK:
        for t in _PINNED_SCALARS:
            if t[0] is buf:
                t[1] = None


def queue_raw_level(warmup, fmt: str, out_slot) -> None:
    """Queue the wideband-level estimate of raw frames (``iqa_raw_level``: mean square of up to 65536 values spread over
    ``warmup``) into ``out_slot`` (double[1], device or pinned host memory) on the current stream."""
    x, n = _as_frames(warmup, fmt)
    flat = x.view(D.torch_mod().float32) if x.is_complex() else x
    N.call("iqa_raw_level", c_int32(P.FMT_CODE[fmt]), N.ptr(flat), c_int64(flat.numel()), N.ptr(out_slot), N.stream_ptr())


def wideband_rms_from(mean_square: float, fmt: str) -> float:
    """RMS of the complex samples (fraction of full scale) from the mean square of the raw values."""
    return math.sqrt(max(2.0 * float(mean_square), 0.0)) * P.INGEST_SCALE[fmt]


#: Precision guard.  The fixed-point channelizers' error is a fraction of the WIDEBAND level whatever the channel holds
#: (tap rounding x wideband RMS, 1..5 x that on tonal captures, plus a level-independent floor: MfmaPlan.z_error_rms), and
#: the FM discriminator divides by the channel's own level: audio error ~ 0.024 x error / |z|.  An NFM channel whose
#: probed level is below guard x (expected z error) is therefore channelized at the next precision that clears it
#: (measured: a -70 dBFS NFM signal beside a full-scale tone comes out 2.8e-4 RMS off the reference at "fast").
PRECISION_GUARD = 1000.0


def pick_precision(kernel_for, base: str, demod_mode: str | None, channel_power, wideband_rms, guard: float | None = None,
                   memo: dict | None = None) -> str:
    """The cheapest precision, not below ``base``, at which a channel of mean power ``channel_power`` (|z|^2, from the
    mixer-sign probe) in a capture of wideband RMS ``wideband_rms`` keeps the 1e-4 audio bar.  Only NFM is guarded (AM
    and SSB do not divide by |z|).  ``kernel_for(precision)`` returns the planned ``_ChannelKernel``; ``memo`` (a dict the
    caller keeps per channel and sign) remembers each precision's (tap-rounding norm, floor), so that a batch asks the
    kernels once, not once per capture."""
    levels = _ChannelKernel.PRECISIONS
    guard = PRECISION_GUARD if guard is None else guard
    if not guard or channel_power is None or wideband_rms is None or (demod_mode or "").lower() not in ("nfm", "fm"):
        return base
    level = math.sqrt(max(float(channel_power), 0.0))
    for name in levels[levels.index(base):]:
        if name == "float32":
            break
        known = None if memo is None else memo.get(name)
        if known is None:
            k = kernel_for(name)
            if k.precision != name:  # (this capture format has no such kernel: uint8 "full", float32 anything)
                known = False
            elif not k._mfma_ok:
                known = (0.0, 0.0)
            else:
                mp = k._ensure_mfma()
                known = (float(mp.err_norm), float(mp.floor_rms))
            if memo is not None:
                memo[name] = known
        if known is False:
            continue
        err = math.hypot(known[0] * wideband_rms, known[1])
        if err <= 0.0 or level >= guard * err:
            return name
    return "float32"


def base_precision(demod_mode: str | None, agc_enabled: bool) -> str:
    """SSB with the AGC on is ill-conditioned in the reference itself: ``_apply_agc`` adds 0.001*(target/|s| - gain) per
    sample for |s| down to 1e-6 (decoders/ssb.py:75-77), so a z difference of 1e-6 near a zero crossing moves the gain
    by hundreds.  Those targets take the "full" precision (z error below the float32 rounding of z itself); everything
    else starts at "fast"."""
    return SSB_AGC_PRECISION if ((demod_mode or "").lower() in ("usb", "lsb", "ssb") and agc_enabled) else "fast"


SSB_AGC_PRECISION = "full"


class MixSignProbe:
    """``choose_mix_sign`` split into an asynchronous launch and a blocking ``result()``, so the
    caller can plan the channelizer while the two probes run."""

    #: iqa_mean_power handles up to this many samples with its single-block, atomics-free kernel
    DIRECT_MAX = 65536

    def __init__(self, warmup, sample_rate: float, freq_offset: float, taps: np.ndarray, decimation: int, *,
                 fmt: str = "f32", iq_order: str = "iq", record_done: bool = True, matrix_cores: bool = True,
                 measure_level: bool = False):
        """``record_done=False``: the caller sets ``_done`` to event(s) of its own that lie behind both probes (an
        event record between two kernels of a stream costs ~7 us on this part).  ``matrix_cores=False``: the probes go
        through the float32 kernel (a few thousand outputs: tens of microseconds), which -- unlike a ring-kernel launch
        -- finds room on a CU beside a running channelizer pass.  ``measure_level``: also estimate the wideband level of
        ``warmup`` (``wideband_rms`` after ``result()`` / ``peek()``: what the precision guard compares ``power`` with)."""
        self._matrix_cores = bool(matrix_cores)
        self._powers = None
        self.power = None
        self.wideband_rms = None
        self._fmt = fmt
        self._level = bool(measure_level)
        self._valid = [False, False]
        x_all, n_in = _as_frames(warmup, fmt)
        if n_in == 0:
            return
        ntaps = len(taps)
        max_len = max(int(sample_rate * 0.05), ntaps * 4, 131_072)
        snippet_len = min(n_in, max_len)
        if snippet_len < ntaps:
            snippet_len = min(n_in, ntaps * 2)
        x = x_all[:snippet_len] if x_all.is_complex() else x_all[: 2 * snippet_len]
        decim = max(decimation, 1)
        self._powers = D.empty(2, "float64")  # iqa_mean_power overwrites its slot
        self._host = _pinned_scalars(id(self))
        self._sign = None
        if not (self._matrix_cores and self._probe_pair(x_all, n_in, snippet_len, taps, sample_rate, freq_offset, decim, fmt, iq_order)):
            for i, sign in enumerate((1, -1)):
                self._probe_one(i, sign, x_all, n_in, x, snippet_len, taps, sample_rate, freq_offset, decim, fmt, iq_order)
        if self._level:
            queue_raw_level(warmup, fmt, self._host[2:3])
        self._done = None
        if record_done:
            self._done = D.torch_mod().cuda.Event()
            self._done.record()

    @classmethod
    def from_powers(cls, host_pair, done_event, level_slot=None, fmt: str = "s16"):
        """A probe whose two mean powers (sign +1, sign -1) are being written into ``host_pair`` (a pinned float64[2] the
        caller owns) by launches already queued; ``done_event`` lies behind them.  See ``probe_targets``."""
        self = cls.__new__(cls)
        self._powers, self.power, self._valid, self._sign = host_pair, None, [True, True], None
        self._host, self._done, self._matrix_cores = host_pair, done_event, True
        self.wideband_rms, self._fmt, self._level, self._level_slot = None, fmt, level_slot is not None, level_slot
        return self

    def _probe_pair(self, x_all, n_in, snippet_len, taps, sample_rate, freq_offset, decim, fmt, iq_order) -> bool:
        """Both signs in two launches instead of four: the two channelizers as the two lanes of one matrix-core launch
        over the snippet (shared ingest) and one reduction with a workgroup per sign, written into the pinned slot.
        Only where ``_probe_one`` would take its interior-only path for both signs and a probe is short enough for the
        single-workgroup reduction; False (nothing queued) otherwise."""
        if fmt not in ("s16", "u8"):
            return False
        ntaps = len(taps)
        n_z = -(-snippet_len // decim)
        discard = min(ntaps, n_z // 4)
        keep = n_z - discard
        if keep <= 0 or keep > self.DIRECT_MAX:
            return False
        chans = [Channelizer(taps, sample_rate=sample_rate, freq_offset=freq_offset, mix_sign=sign, decimation=decim, fmt=fmt,
                             iq_order=iq_order) for sign in (1, -1)]
        z_keep = D.empty(2 * keep, "complex64")
        if not ChannelBank(chans).run_interior_only(x_all, n_in, discard, keep, [z_keep[:keep], z_keep[keep:]]):
            return False
        N.call("iqa_mean_power_batch", N.ptr(z_keep), c_int64(keep), c_int32(2), c_int64(0), N.ptr(self._host), N.stream_ptr())
        self._valid = [True, True]
        return True

    def _probe_one(self, i, sign, x_all, n_in, x, snippet_len, taps, sample_rate, freq_offset, decim, fmt, iq_order):
        """Queue the probe for one mixer sign on the current stream; its mean power ends up in ``_host[i]``."""
        ntaps = len(taps)
        ch = Channelizer(taps, sample_rate=sample_rate, freq_offset=freq_offset, mix_sign=sign, decimation=decim,
                         fmt=fmt, iq_order=iq_order)
        n_z = -(-snippet_len // decim)
        discard = min(ntaps, n_z // 4)
        if n_z - discard == 0:
            discard = 0
        # Only z[discard:] enters the power (processing.py:651-656), and those outputs see neither the zero
        # initial state nor anything past the snippet: when the warm-up buffer is longer than the snippet they are
        # all interior outputs of the matrix-core kernel -- one launch per sign instead of three.
        z_keep = D.empty(n_z - discard, "complex64")
        if self._matrix_cores and fmt in ("s16", "u8") and ch._kernel.run_interior_only(x_all, n_in, discard, n_z - discard, z_keep):
            # a short reduction is one block that WRITES its result: straight into the mapped pinned slot, no
            # device scalar and no copy behind it (a blit between kernels costs ~13 us of a 0.9 ms capture)
            direct = z_keep.numel() <= self.DIRECT_MAX
            _mean_power_into(z_keep, 0, (self._host if direct else self._powers)[i : i + 1])
            self._valid[i] = True
            if not direct:
                self._host[i : i + 1].copy_(self._powers[i : i + 1], non_blocking=True)
            return
        z = ch.process(x, last_block=True)
        if z.numel():
            discard = min(ntaps, z.numel() // 4)
            if z.numel() - discard == 0:
                discard = 0
            direct = z.numel() - discard <= self.DIRECT_MAX  # (one workgroup writes the mean: straight into the pinned slot)
            _mean_power_into(z, discard, (self._host if direct else self._powers)[i : i + 1])
            self._valid[i] = True
            if not direct:
                self._host[i : i + 1].copy_(self._powers[i : i + 1], non_blocking=True)

    def result(self) -> int:
        if self._powers is None:
            return 1
        if self._sign is None:
            self._wait()
            self._sign = self._decide()
            _release_scalars(self._host)
            self._host = None
        return self._sign

    def _decide(self) -> int:
        host = self._host.numpy()
        best_sign, best_power = 1, -np.inf
        for i, sign in enumerate((1, -1)):
            power = float(host[i]) if self._valid[i] else -np.inf
            if power > best_power:
                best_power, best_sign = power, sign
        self.power = best_power if np.isfinite(best_power) else None  # mean |z|^2 of the chosen sign's probe
        if self._level:
            slot = getattr(self, "_level_slot", None)
            self.wideband_rms = wideband_rms_from(float(slot[0]) if slot is not None else float(host[2]), self._fmt)
        return best_sign

    def peek(self) -> int:
        """The sign the two powers in the pinned slot say NOW, without giving the slot back: for probes that live inside
        a captured graph (every replay writes the same slot again; the caller has waited for the replay)."""
        return 1 if self._powers is None else self._decide()

    def _wait(self) -> None:
        done = self._done
        for ev in (done if isinstance(done, (list, tuple)) else [done]):
            ev.synchronize()

    def __del__(self):
        try:
            if getattr(self, "_host", None) is not None and self._sign is None:
                self._wait()  # the write into the pinned buffer must not land in someone else's read-back
                _release_scalars(self._host)
        except Exception:
            pass


def choose_mix_sign(warmup, sample_rate: float, freq_offset: float, taps: np.ndarray, decimation: int, *,
                    fmt: str = "f32", iq_order: str = "iq") -> int:
    """Pick the mixer sign that puts more power in the channel (reference processing.py:623-663).

    For each sign the first ``min(len, max(0.05*fs, 4L, 131072))`` samples are mixed, filtered
    from zero state, decimated (``[::D]``) and the mean power after dropping the first
    ``min(L, n/4)`` decimated samples is compared; strictly greater wins, ties give +1.
    ``warmup`` is complex64 (NumPy / tensor) or raw frames when ``fmt`` is 's16'/'u8'.
    Both probes are enqueued before the single host read-back of the two powers.
    """
    return MixSignProbe(warmup, sample_rate, freq_offset, taps, decimation, fmt=fmt, iq_order=iq_order).result()


# --------------------------------------------------------------------------------------------- #
# pipeline                                                                                      #
# --------------------------------------------------------------------------------------------- #


@dataclass
class ProcessingResult:
    """reference processing.py:666-675"""

    sample_rate_probe: SampleRateProbe
    center_freq: float
    target_freq: float
    freq_offset: float
    decimation: int
    fs_channel: float
    mix_sign: int
    audio_peak: float


class ProcessingCancelled(RuntimeError):  # noqa: N818
    """Raised when processing is aborted early by user request (reference processing.py:678)."""


class ChannelDemod:
    """Demodulate + AudioWriter.write for many reference chunks in one fused call.

    ``decoder.process`` (processing.py:1128) followed by ``audio_writer.write`` (:1147) for a whole
    block: the only per-chunk behaviour of the reference decoders is the SSB AGC restart
    (decoders/ssb.py:72), expressed as restart indices of the segmented scan; the writer's pre-clip
    peak, +-0.99 clip and the per-chunk sum of squares (rms_dbfs) are fused into the last scan pass
    (``iqa_demodulate``).  ``self.decoder`` is the matching pluggable decoder object (kept for its
    parameters and API parity; the fused path carries its own device state).
    """

    def __init__(self, mode: str, fs_channel: float, *, deemph_us: float, agc_enabled: bool):
        self.decoder = create_decoder(mode, deemph_us=deemph_us, agc_enabled=agc_enabled)
        self.decoder.setup(fs_channel)
        self.params = self.decoder.fused_params()
        self._needs_scratch = self.params.mode in (N.DEMOD_MODE["usb"], N.DEMOD_MODE["lsb"]) and bool(self.params.agc_enabled)
        self.chunk_sumsq: list = []  # (device float64[n_chunks*8], counts)
        self._blk = None  # one device block: [state 32 B | peak 4 B (+pad to 64) | sumsq n_chunks*8 f64]
        self._starts_key = None
        self._fresh = False
        self._from_reset = False
        self._alloc_block(0)

    # {float2 prev = 1+0j; double y_last; double x_last, y_last} -- the states of a decoder that has seen nothing
    _STATE0 = np.array([1.0, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32)

    def _alloc_block(self, n_chunks: int) -> None:
        torch = D.torch_mod()
        img = np.zeros(64 + n_chunks * 64, dtype=np.uint8)
        img[:32] = self._STATE0.view(np.uint8)
        self._img, self._img_host = img, None
        self._blk = D.from_numpy(img)
        self._n_chunks = n_chunks
        self.state_dev = self._blk[:32].view(torch.float32)
        self.peak_dev = self._blk[32:36].view(torch.float32)
        self._sumsq = self._blk[64:].view(torch.float64)
        self._fresh = True

    def reset(self, force: bool = False) -> None:
        """Back to a decoder that has seen nothing (states, peak, per-chunk sums): one small H2D copy from a
        pinned image (pinned on the first reset: pin_memory() costs milliseconds)."""
        self.chunk_sumsq = []
        if self._fresh and not force:  # (force: a step being captured into a graph must not depend on what ran before it)
            return
        # nothing is copied: the next ``process`` starts from the initial state by itself and clears the peak and the
        # per-chunk sums (iqa_demodulate_from_reset) -- one node less per capture in a captured step
        self._from_reset = True
        self._fresh = True

    def prepare(self, n: int, chunk_starts: np.ndarray):
        """Upload / allocate everything ``process`` needs for a block of ``n`` samples ahead of time, so the
        launches that follow a long kernel are not preceded by host-side copies.  The chunk starts and the scan
        workspace are kept while the block shape stays the same (a batch of equal captures uploads them once)."""
        starts64 = np.ascontiguousarray(chunk_starts, dtype=np.int64)
        key = (n, len(starts64), hash(starts64.tobytes()))
        if key != self._starts_key:
            self._starts_dev = D.from_numpy(starts64)
            self._work = D.empty(int(N.lib().iqa_scan_workspace_bytes(n)), "uint8")
            self._scratch = D.empty(n, "float32") if self._needs_scratch else None
            self._starts_key = key
        if self._fresh and not self.chunk_sumsq:
            if len(starts64) != self._n_chunks:
                self._alloc_block(len(starts64))  # nothing processed yet: the block is simply re-made at this size
            sumsq = self._sumsq
        else:  # later blocks of a streaming run: their own sums (all read back at the end)
            sumsq = D.zeros(len(starts64) * 8, "float64")  # IQA_SUMSQ_SLOTS sub-slots per chunk
        self._prepared = (n, len(starts64), self._starts_dev, sumsq, self._work, self._scratch)

    def process(self, z_dev, chunk_starts: np.ndarray, out_dev):
        """z_dev -> clipped float32 audio written into ``out_dev`` (len == len(z_dev))."""
        n = int(z_dev.numel())
        if n == 0:
            return
        prep = getattr(self, "_prepared", None)
        if prep is None or prep[0] != n or prep[1] != len(chunk_starts):
            self.prepare(n, chunk_starts)
        _, _, starts_dev, sumsq, work, scratch = self._prepared
        self._prepared = None
        self._fresh = False
        entry = "iqa_demodulate_from_reset" if self._from_reset else "iqa_demodulate"
        self._from_reset = False
        N.call(entry, byref(self.params), N.ptr(z_dev), c_int64(n), N.ptr(self.state_dev), N.ptr(starts_dev),
               c_int64(len(chunk_starts)), N.ptr(self.peak_dev), N.ptr(sumsq), N.ptr(out_dev), N.ptr(scratch), N.ptr(work),
               N.stream_ptr())
        counts = np.diff(np.append(chunk_starts, n))
        self.chunk_sumsq.append((sumsq, counts))

    @property
    def peak(self) -> float:
        return 0.0 if self._from_reset else float(self.peak_dev.item())  # (a reset not yet followed by a block)

    def chunk_rms_dbfs(self) -> list[float]:
        out = []
        for sumsq, counts in self.chunk_sumsq:
            for s, c in zip(sumsq.cpu().numpy().reshape(-1, 8).sum(axis=1), counts):
                if c > 0:
                    out.append(20.0 * math.log10(math.sqrt(float(s) / float(c) + 1e-18) + 1e-12))
        return out


class Resampler48k:
    """The ``-ar 48000 -acodec pcm_s16le`` leg (reference processing.py:399-418) on the GPU.
    Build-defined specification (dsp_plan.plan_resampler); parity with libswresample is unpinned."""

    _TABLES: dict = {}  # (device, declared input rate) -> the polyphase table on that device (read-only, 12.9 MB at 96 154 Hz)

    def __init__(self, fs_channel: float):
        self.plan = P.plan_resampler(fs_channel)
        key = (D.torch_mod().cuda.current_device(), self.plan.in_rate)
        with _KERNEL_CACHE_LOCK:
            table = self._TABLES.get(key)
        if table is None:
            table = D.from_numpy(self.plan.table.reshape(-1))
            with _KERNEL_CACHE_LOCK:
                if len(self._TABLES) >= 16:
                    self._TABLES.clear()
                self._TABLES[key] = table
        self.table_dev = table

    def process(self, audio_dev, *, want: str = "f32"):
        """48 kHz audio of a whole stream: float32 (``want="f32"``), PCM16 (``"pcm16"``: what the WAV holds, rounded
        from the float32 value in the same pass) or the pair (``"both"``)."""
        if want not in ("f32", "pcm16", "both"):
            raise ValueError(f"want must be 'f32', 'pcm16' or 'both', not {want!r}")
        n_in = int(audio_dev.numel())
        n_out = self.plan.n_out(n_in)
        if self.plan.up == 1 and self.plan.down == 1:
            # a channel rate of exactly 48 kHz: `-ar 48000` on a 48 kHz stream resamples nothing (the build-defined
            # specification says so too: oracle resample_48k) -- the stream itself, and its PCM16
            y = audio_dev.clone() if want != "pcm16" else None
            pcm = self.to_pcm16(audio_dev) if want != "f32" else None
            return y if want == "f32" else pcm if want == "pcm16" else (y, pcm)
        y = D.empty(n_out, "float32") if want != "pcm16" else None
        pcm = D.empty(n_out, "int16") if want != "f32" else None
        if n_out:
            N.call("iqa_resample", N.ptr(audio_dev), c_int64(n_in), N.ptr(self.table_dev), c_int32(self.plan.up),
                   c_int32(self.plan.down), c_int32(self.plan.half_taps), c_int64(0), c_int64(n_out), N.ptr(y), N.ptr(pcm),
                   N.stream_ptr())
        return y if want == "f32" else pcm if want == "pcm16" else (y, pcm)

    @staticmethod
    def to_pcm16(y_dev):
        pcm = D.empty(y_dev.numel(), "int16")
        if y_dev.numel():
            N.call("iqa_float_to_pcm16", N.ptr(y_dev), c_int64(y_dev.numel()), N.ptr(pcm), N.stream_ptr())
        return pcm


class _BlockStager:
    """Capture blocks: memory-mapped file -> one of two pinned host buffers (filled by a helper thread,
    overlapping the previous block's GPU work) -> device tensor by asynchronous H2D copy.
    Replaces the reference's ffmpeg decode pipe + ``IQReader.read_block`` (processing.py:238-266): the
    capture's own sample format goes to the GPU untouched."""

    #: pinned staging buffers that finished runs gave back, by (dtype, elements): pinning 50 MB costs ~15 ms and a run
    #: on a small capture (the reference's --benchmark: 50 MB) spent a third of its wall time on it
    _POOL: dict = {}
    _POOL_MAX_BYTES = 2 << 30
    _POOL_LOCK = threading.Lock()

    def __init__(self, frames: np.ndarray, dtype: str, max_frames: int):
        torch = D.torch_mod()
        self.frames = frames
        self.dtype = getattr(torch, dtype)
        self.numel = 2 * max_frames
        self.bufs = [None, None]  # pinned lazily: a capture of one block needs one
        self.events = [None, None]
        self.pending = None  # (thread, slot, lo, hi)
        self.slot = 0

    def _buf(self, slot: int):
        if self.bufs[slot] is None:
            torch = D.torch_mod()
            key = (str(self.dtype), self.numel)
            with self._POOL_LOCK:
                stack = self._POOL.get(key)
                self.bufs[slot] = stack.pop() if stack else None
            if self.bufs[slot] is None:
                self.bufs[slot] = torch.empty(self.numel, dtype=self.dtype, pin_memory=True)
        return self.bufs[slot]

    def close(self) -> None:
        """Give the pinned buffers back (every H2D copy out of them has completed)."""
        if self.pending is not None:
            self.pending[0].join()
            self.pending = None
        for ev in self.events:
            if ev is not None:
                ev.synchronize()
        key = (str(self.dtype), self.numel)
        with self._POOL_LOCK:
            held = sum(t.numel() * t.element_size() for st in self._POOL.values() for t in st)
            for i, b in enumerate(self.bufs):
                if b is not None and held + b.numel() * b.element_size() <= self._POOL_MAX_BYTES:
                    self._POOL.setdefault(key, []).append(b)
                    held += b.numel() * b.element_size()
                self.bufs[i] = None

    #: helper threads of one block copy (file mapping -> pinned buffer): a single memcpy moves ~2.7 GB/s on the GPU boxes'
    #: hosts, which made the FILE the bottleneck of a file -> WAV run (10 s @ 10 MS/s: 0.148 s, of which ~0.1 s this copy)
    copy_threads = 8

    def _fill(self, slot: int, lo: int, hi: int) -> None:
        import threading

        dst = self._buf(slot).numpy()[: 2 * (hi - lo)]
        src = self.frames[2 * lo : 2 * hi]
        parts = min(int(self.copy_threads), max(1, dst.size // (1 << 22)))  # (NumPy copies release the GIL)
        if parts <= 1:
            np.copyto(dst, src)
            return
        cuts = [dst.size * i // parts for i in range(parts + 1)]
        workers = [threading.Thread(target=np.copyto, args=(dst[a:b], src[a:b]), name="iq-stager-copy", daemon=True)
                   for a, b in zip(cuts[1:-1], cuts[2:])]
        for w in workers:
            w.start()
        np.copyto(dst[: cuts[1]], src[: cuts[1]])
        for w in workers:
            w.join()

    def prefetch(self, lo: int, hi: int) -> None:
        import threading

        slot = self.slot ^ 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # the previous H2D out of this buffer must be done
        th = threading.Thread(target=self._fill, args=(slot, lo, hi), name="iq-stager", daemon=True)
        th.start()
        self.pending = (th, slot, lo, hi)

    def fetch(self, lo: int, hi: int):
        torch = D.torch_mod()
        if self.pending is not None and self.pending[2:] == (lo, hi):
            th, slot = self.pending[0], self.pending[1]
            th.join()
        else:
            slot = self.slot ^ 1
            if self.events[slot] is not None:
                self.events[slot].synchronize()
            self._fill(slot, lo, hi)
        self.pending = None
        self.slot = slot
        dev = torch.empty(2 * (hi - lo), dtype=self.dtype, device=D.device())
        dev.copy_(self._buf(slot)[: 2 * (hi - lo)], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev
        return dev


class ProcessingPipeline:
    """``ProcessingPipeline(config).run(progress_sink) -> ProcessingResult``, ``.cancel()``
    (reference processing.py:682-1213).

    Differences that are deliberate and documented in DESIGN.md: the capture is memory-mapped
    and streamed to HBM in blocks of whole reference chunks (``block_chunks``); one kernel does
    ingest+mix+filter+decimate; the 48 kHz resample/PCM16 encode runs on the GPU at the end;
    no ffmpeg/ffprobe processes exist.  Per-chunk semantics (AGC restart points, mix-sign
    warm-up window, sample counts) are the reference's.
    """

    #: frames per device block (rounded down to whole chunks); 64 Mi frames = 256 MiB of int16 I/Q
    block_frames_target = 64 * 1024 * 1024

    def __init__(self, config: ProcessingConfig):
        self.config = config
        self._cancelled = False
        self._resolved_chunk_size: int | None = None
        self.chunk_rms_dbfs: list[float] = []
        self.audio_fs_channel = None  # device float32 tensor of the clipped channel-rate audio (kept for tests)
        self.keep_channel_audio = False
        self.channelizer_kernel = None  # name of the channelizer kernel that produced the last block of the run
        self.f32_integer_path = True  # float32 captures whose values are all k / 32768 run as int16 on the matrix cores
        self.channelizer_precision = None  # "fast" / "fine" / "full" / "float32": what the run's channelizer was planned at

    def cancel(self) -> None:
        self._cancelled = True
        if getattr(self, "_multi", None) is not None:
            self._multi.cancel()

    def _is_pass_through_mode(self) -> bool:
        return (self.config.demod_mode or "").lower() in {"none", "pass", "iq"}

    def _effective_chunk_size(self, sample_rate: float) -> int:
        if self._resolved_chunk_size is None:
            self._resolved_chunk_size = tune_chunk_size(sample_rate, self.config.chunk_size)
        return self._resolved_chunk_size

    def _default_output_path(self, info: iqio.CaptureInfo) -> Path:
        ft = int(self.config.target_freq)
        if self._is_pass_through_mode():
            suffix = self.config.in_path.suffix
            if info.container == "wav":
                ext = suffix if suffix.lower() in {".wav", ".wave", ".wv", ".rf64"} else ".wav"
            else:
                ext = suffix or {"pcm_u8": ".cu8", "pcm_s16le": ".cs16", "pcm_f32le": ".cf32"}.get(info.codec, ".raw")
            return self.config.in_path.with_name(f"slice_{ft}{ext}")
        return self.config.in_path.with_name(f"audio_{ft}_48k.wav")

    def run(self, progress_sink: ProgressSink | None = None) -> ProcessingResult:
        """One target frequency: a :class:`MultiChannelPipeline` with a single channel."""
        multi = MultiChannelPipeline([self.config], _owner=self)
        self._multi = multi
        if self._cancelled:
            multi.cancel()
        return multi.run(progress_sink)[0]


class _Target:
    """Per-target streaming state of a run: channelizer, demodulator, output buffers."""

    def __init__(self, cfg: ProcessingConfig, owner, *, info, sample_rate, center_freq, decimation, fs_channel, total):
        self.cfg, self.owner, self.info = cfg, owner, info
        self.sample_rate, self.decimation, self.fs_channel, self.total = sample_rate, decimation, fs_channel, total
        self.center_freq = center_freq
        self.target_freq = cfg.target_freq if cfg.target_freq > 0 else center_freq
        self.freq_offset = self.target_freq - center_freq
        self.pass_through = (cfg.demod_mode or "").lower() in {"none", "pass", "iq"}
        self.taps = design_channel_filter(sample_rate, cfg.bandwidth, decimation)
        LOG.info("Designed FIR channel filter with %d taps.", len(self.taps))
        if cfg.filter_block <= 0:
            raise ValueError("block_size must be positive")
        self.demod = None if self.pass_through else ChannelDemod(cfg.demod_mode, fs_channel, deemph_us=cfg.deemph_us,
                                                                 agc_enabled=cfg.agc_enabled)
        if cfg.iq_order not in N.ORDER:
            raise ValueError(f"Unsupported iq_order '{cfg.iq_order}'")
        self.chan = None
        self.sign_probe = None
        self.mix_sign = 1
        self.pos_dec = 0
        self.peak = 0.0
        self.output_path = cfg.output_path if cfg.output_path else owner._default_output_path(info)

    def _channelizer(self, sign: int, exact: bool = False, fmt: str | None = None, precision: str | None = None) -> Channelizer:
        return Channelizer(self.taps, sample_rate=self.sample_rate, freq_offset=self.freq_offset, mix_sign=sign,
                           decimation=self.decimation, fmt=fmt or self.info.fmt, iq_order=self.cfg.iq_order, exact=exact,
                           precision=precision)

    #: Precision guard (see ``pick_precision``); 0 switches it off.
    precision_guard = PRECISION_GUARD

    def _pick_precision(self, probe_power, wideband_rms) -> str:
        """The precision this target's channelizer runs at: "full" for SSB with the AGC on, otherwise "fast" unless the
        precision guard asks for more.  A float32 capture that may run as int16 (``f32_integer_path``) is judged by its
        int16 twin -- that is the kernel its blocks take."""
        base = "fast" if self.demod is None else base_precision(self.cfg.demod_mode, self.cfg.agc_enabled)
        fmt = self.info.fmt
        if fmt == "f32" and getattr(self.owner, "f32_integer_path", False):
            fmt = "s16"
        return pick_precision(lambda name: self._channelizer(self.mix_sign, fmt=fmt, precision=name)._kernel, base,
                              None if self.demod is None else self.cfg.demod_mode, probe_power, wideband_rms, self.precision_guard)

    def begin(self, warm) -> None:
        """Launch the mixer-sign probes (asynchronously) and plan the channelizer for the likely sign meanwhile."""
        if self.cfg.mix_sign_override in (1, -1):
            self.mix_sign = self.cfg.mix_sign_override
            self.chan = self._channelizer(self.mix_sign)
        else:
            self.sign_probe = MixSignProbe(warm, self.sample_rate, self.freq_offset, self.taps, self.decimation,
                                           fmt=self.info.fmt, iq_order=self.cfg.iq_order)
            self.chan = self._channelizer(1)
        self.precision = "fast"

    def settle(self, wideband_rms=None) -> None:
        power = None
        if self.sign_probe is not None:
            self.mix_sign = self.sign_probe.result()
            power = self.sign_probe.power
            self.sign_probe = None
        self.precision = self._pick_precision(power, wideband_rms)
        if self.precision != "fast" and (self.cfg.demod_mode or "").lower() in ("nfm", "fm"):
            LOG.info("Channel level %.1f dBFS against a wideband level of %.1f dBFS: '%s' channelizer for this target.",
                     10.0 * math.log10(max(power or 0.0, 1e-30)), 20.0 * math.log10(max(wideband_rms or 0.0, 1e-15)), self.precision)
        self.chan = self._channelizer(self.mix_sign, precision=self.precision)
        self.chan.plan_ahead()
        self.owner.channelizer_precision = self.precision  # (the decision; a float32 capture's int16 twin runs at it)
        LOG.info("Selected mixer sign %d based on warm-up snippet.", self.mix_sign)
        n_dec_total = -(-self.total // self.decimation)
        self.z_all = D.empty(n_dec_total, "complex64") if (self.pass_through or self.cfg.dump_iq_path) else None
        self.audio_all = None if self.pass_through else D.empty(n_dec_total, "float32")

    def before_block(self, done: int, n: int, chunk: int) -> None:
        """Upload what the demodulator needs for the block of ``n`` frames at frame ``done`` ahead of the channelizer."""
        _, n_out = self.chan.outputs_for(n)
        self._starts = None
        if self.demod is not None and n_out:
            self._starts = P.chunk_output_starts(chunk, self.decimation, done, n)
            self.demod.prepare(n_out, self._starts)

    def after_block(self, z, tracker) -> None:
        """The block's decimated stream ``z`` -> dump / slice buffers, demodulator, progress."""
        n_out = int(z.numel())
        tracker.advance("channel", float(n_out))
        if self.z_all is not None and n_out:
            self.z_all[self.pos_dec : self.pos_dec + n_out] = z
            if self.cfg.dump_iq_path:
                tracker.advance("dump_iq", float(n_out))
        if self.demod is not None and n_out:
            self.demod.process(z, self._starts, self.audio_all[self.pos_dec : self.pos_dec + n_out])
        tracker.advance("demod", float(n_out))
        tracker.advance("encode", n_out / max(self.fs_channel, 1e-9) * 48_000.0)
        self.pos_dec += n_out

    def finish(self) -> None:
        cfg, info = self.cfg, self.info
        self.owner.channelizer_kernel = getattr(self, "chan16_kernel", None) or self.chan._kernel.last_kernel
        self.output_path.parent.mkdir(parents=True, exist_ok=True)
        if cfg.dump_iq_path:
            Path(cfg.dump_iq_path).write_bytes(self.z_all[: self.pos_dec].cpu().numpy().astype(np.complex64).tobytes())
        if self.pass_through:
            zs = self.z_all[: self.pos_dec].cpu().numpy()
            self.peak = float(np.max(np.abs(zs))) if zs.size else 0.0
            values = iqio.encode_iq_slice(zs, info.codec, info.container)
            if info.container == "wav":
                iqio.write_wav_iq(self.output_path, values, max(1, int(round(self.fs_channel))), info.fmt)
            else:
                self.output_path.write_bytes(values.tobytes())
            return
        self.demod.decoder.finalize()
        audio = self.audio_all[: self.pos_dec]
        if self.owner.keep_channel_audio:
            self.owner.audio_fs_channel = audio
        rs = Resampler48k(self.fs_channel)
        pcm = rs.process(audio, want="pcm16").cpu().numpy()
        iqio.write_wav_pcm16(self.output_path, pcm, 48_000)
        self.peak = self.demod.peak
        self.owner.chunk_rms_dbfs = self.demod.chunk_rms_dbfs()
        LOG.info("Audio peak level %.2f dBFS.", 20.0 * math.log10(max(self.peak, 1e-6)))


class MultiChannelPipeline:
    """Several target frequencies of ONE capture in a single pass over the file.

    The reference CLI runs a whole pipeline per ``--ft`` target, re-decoding the input each time
    (cli.py:683-710).  Here every block of the capture is staged and uploaded once, and the channels that share a
    decimation are extracted by ONE launch of the channelizer over the HBM-resident block (:class:`ChannelBank`: every
    (channel, tap-row group) is a lane of the ring kernel; BASELINE configs 3/5); each channel keeps exactly the
    per-target semantics of :class:`ProcessingPipeline` (own mixer-sign probe, taps, decoder, output).
    ``configs`` must agree on the input file and its interpretation.
    """

    def __init__(self, configs: list, _owner=None):
        if not configs:
            raise ValueError("at least one ProcessingConfig is required")
        if len(configs) > 5 and _owner is None:
            LOG.debug("more than the reference CLI's five targets in one pass (%d)", len(configs))
        first = configs[0]
        for c in configs[1:]:
            same = (c.in_path == first.in_path and c.input_format == first.input_format and c.input_container == first.input_container
                    and c.input_sample_rate == first.input_sample_rate and c.chunk_size == first.chunk_size
                    and c.max_input_seconds == first.max_input_seconds and c.center_freq == first.center_freq)
            if not same:
                raise ValueError("all targets of a multi-channel run must share the input file, format, rate and chunking")
        self.configs = configs
        self.owners = [_owner] if _owner is not None else [ProcessingPipeline(c) for c in configs]
        self._cancelled = False

    def cancel(self) -> None:
        self._cancelled = True

    def run(self, progress_sink: ProgressSink | None = None) -> list:
        cfg = self.configs[0]
        tracker = ProgressTracker(progress_sink)
        targets: list[_Target] = []

        def _request_cancel() -> None:
            self._cancelled = True
            tracker.cancel()
            tracker.status("Cancelling…")

        def _check_cancel(stage: str = "") -> None:
            if self._cancelled or tracker.cancelled or any(o._cancelled for o in self.owners):
                self._cancelled = True
                LOG.info("Processing cancelled during %s.", stage or "run")
                raise ProcessingCancelled("Processing cancelled by user.")

        if progress_sink is not None:
            with contextlib.suppress(AttributeError):
                progress_sink.set_cancel_callback(_request_cancel)

        manual_rate = cfg.input_sample_rate
        if manual_rate is not None and manual_rate <= 0:
            raise ValueError("Input sample rate override must be positive.")
        try:
            info = iqio.probe_capture(cfg.in_path, input_format=cfg.input_format, input_container=cfg.input_container,
                                      input_sample_rate=manual_rate)
            for c in self.configs:
                c.input_container = c.input_container or info.container
                c.input_format = c.input_format or info.codec
            if info.container == "raw" and manual_rate is None:
                raise ValueError("Raw IQ inputs require --input-sample-rate (CLI) or a manual entry in the GUI.")
            if info.sample_rate is None or info.sample_rate <= 0:
                raise RuntimeError("Unable to determine input sample rate automatically. Provide --input-sample-rate.")
            sample_rate = float(info.sample_rate)
            rate_probe = SampleRateProbe(header=None if manual_rate else sample_rate, wave=sample_rate)

            preview_seconds = cfg.max_input_seconds
            if preview_seconds is not None and preview_seconds <= 0:
                preview_seconds = None
            total = info.n_frames
            if preview_seconds is not None:
                total = min(total, max(1, int(math.floor(preview_seconds * sample_rate))))

            for c in self.configs:
                if c.target_freq <= 0 and not c.probe_only:
                    raise ValueError("Target frequency must be positive. Provide --ft or use --interactive.")
                if c.bandwidth <= 0:
                    raise ValueError("Bandwidth must be positive.")
            center_freq = cfg.center_freq
            if center_freq is None:
                center_freq, source = iqio.center_frequency_from_filename(cfg.in_path)
                if center_freq is None:
                    raise ValueError(
                        "Center frequency not supplied and could not be determined from metadata or filename. "
                        "Use --fc to provide it explicitly.")
                for c in self.configs:
                    c.center_freq, c.center_freq_source = center_freq, source
            chunk = tune_chunk_size(sample_rate, cfg.chunk_size)
            for o in self.owners:
                o._resolved_chunk_size = chunk

            n_est = 0.0
            for c, o in zip(self.configs, self.owners):
                decimation, fs_channel = P.choose_decimation(sample_rate, c.fs_ch_target)
                LOG.info("Input sample rate %.2f Hz; centre %.0f Hz, target %.0f Hz; decimation %d -> %.2f Hz",
                         sample_rate, center_freq, c.target_freq, decimation, fs_channel)
                n_est += total / max(decimation, 1)
            phases = [PhaseState("ingest", "Ingest IQ", float(total)), PhaseState("channel", "Channelize", n_est),
                      PhaseState("demod", "Demodulate", n_est),
                      PhaseState("encode", "Encode Audio", len(self.configs) * total / sample_rate * 48_000.0)]
            if any(c.dump_iq_path for c in self.configs):
                phases.insert(3, PhaseState("dump_iq", "Write IQ Dump", n_est))
            tracker.start(phases)
            tracker.status("design filter")
            _check_cancel("initialization")
            for c, o in zip(self.configs, self.owners):
                decimation, fs_channel = P.choose_decimation(sample_rate, c.fs_ch_target)
                targets.append(_Target(c, o, info=info, sample_rate=sample_rate, center_freq=center_freq,
                                       decimation=decimation, fs_channel=fs_channel, total=total))
            if total == 0:
                raise RuntimeError("Input stream produced no samples.")

            frames = iqio.map_frames(info)
            np_dt = {"s16": "int16", "u8": "uint8", "f32": "float32"}[info.fmt]
            block = max(1, self.owners[0].block_frames_target // chunk) * chunk
            stager = _BlockStager(frames, np_dt, min(block, total))
            warm = stager.fetch(0, min(chunk, total))
            _check_cancel("warm-up")
            for t in targets:  # all probes are enqueued before the first read-back
                t.begin(warm)
            # wideband level of the warm-up block as a fraction of full scale (for the precision guard): iqa_raw_level, the
            # same estimate the batch runners take (eight stretches spread over the block)
            level = D.zeros(1, "float64")
            queue_raw_level(warm, info.fmt, level)
            wideband_rms = wideband_rms_from(float(level.item()), info.fmt)
            for t in targets:
                t.settle(wideband_rms)
            if cfg.probe_only:
                tracker.advance("ingest", float(warm.numel() // 2))
                return [ProcessingResult(rate_probe, center_freq, t.target_freq, t.freq_offset, t.decimation, t.fs_channel,
                                         t.mix_sign, 0.0) for t in targets]

            # channels that share a decimation share their pass over every block (ChannelBank); built after settle(),
            # which may have replaced a channelizer by the one for the other mixer sign
            banks = []
            for key in sorted({t.decimation for t in targets}):
                members = [t for t in targets if t.decimation == key]  # (a bank runs its "full" / "float32" members one by one)
                banks.append((ChannelBank([t.chan for t in members]), members))
            self.banks = [b for b, _ in banks]
            # float32 captures that are integer captures in disguise (every value k / 32768: what SDR software writes for
            # int16 / 12-bit / int8 ADC samples): each block is re-packed to int16 on the device, checked value by value
            # (iqa_f32_to_s16_exact), and takes the matrix-core channelizers; the first block that is NOT of that form
            # switches the rest of the run to the float32 kernel (whose state has been carried along all the time).
            # Every other float32 capture within the planes' range (RTL-SDR's (u - 127.5) / 127.5, k / 32767, resampled or
            # filtered recordings) is TWO int16 planes, x = 2^shift (hi + lo / 32768) / 32768 exactly to 2^(shift - 31)
            # (iqa_f32_split_s16), and the filter is linear: z = 2^shift (z(hi) + 2^-15 z(lo)), two passes of the int16
            # channelizers (the second at "fast": its input is 2^-15 of the first's).
            banks16 = banks16_lo = None
            shift16 = 0
            if info.fmt == "f32" and self.owners[0].f32_integer_path:
                banks16 = [(ChannelBank([t._channelizer(t.mix_sign, precision=t.precision, fmt="s16") for t in members]), members)
                           for _, members in banks]
                banks16_lo = [(ChannelBank([t._channelizer(t.mix_sign, precision="fast", fmt="s16") for t in members]), members)
                              for _, members in banks]
                flag16 = D.zeros(1, "int32")
                # headroom from the warm-up block's largest value (a later block that exceeds it falls back to the float32 kernel)
                top = float(warm.view(D.torch_mod().float32).abs().max().item())
                while shift16 < 8 and top * 2.0 ** (15 - shift16) > 32767.0:
                    shift16 += 1
            self.integer_blocks = 0  # blocks of a float32 capture that ran as int16 (one plane)
            self.split_blocks = 0  # blocks of a float32 capture that ran as two int16 planes
            done = 0
            while done < total:
                _check_cancel(f"block at frame {done}")
                hi = min(done + block, total)
                raw = warm if (done == 0 and hi <= warm.numel() // 2) else stager.fetch(done, hi)
                if hi < total:
                    stager.prefetch(hi, min(hi + block, total))  # disk -> pinned memory while the GPU works
                n = hi - done
                tracker.advance("ingest", float(n))
                tracker.status(f"channel @ {done}")
                raw16 = raw16_lo = None
                lo_needed = False
                if banks16 is not None:
                    raw16, raw16_lo = D.empty(2 * n, "int16"), D.empty(2 * n, "int16")
                    flag16.zero_()
                    N.call("iqa_f32_split_s16", N.ptr(raw), c_int64(2 * n), c_int32(shift16), N.ptr(raw16), N.ptr(raw16_lo), N.ptr(flag16),
                           N.stream_ptr())
                    bits = int(flag16.item())  # (a host read per 64 Mi-frame block)
                    if bits & 1:
                        LOG.info("float32 capture leaves the range of the int16 planes (shift %d) at frame %d: float32 channelizer.",
                                 shift16, done)
                        banks16 = banks16_lo = raw16 = raw16_lo = None
                    else:
                        lo_needed = bool(bits & 2)
                for bi, (bank, members) in enumerate(banks):  # one pass over the block per decimation, all its channels at once
                    for t in members:
                        t.before_block(done, n, chunk)
                    if raw16 is not None:
                        zs = banks16[bi][0].process(raw16)
                        if lo_needed:
                            zs_lo = banks16_lo[bi][0].process(raw16_lo)
                            zs = [z.add_(zl, alpha=2.0 ** -15) for z, zl in zip(zs, zs_lo)]
                        else:  # (nothing in the low plane of this block: its channelizers' history and position move along)
                            for c in banks16_lo[bi][0].chans:
                                c._advance(raw16_lo, n)
                        if shift16:
                            zs = [z.mul_(2.0 ** shift16) for z in zs]
                        x32, _ = _as_frames(raw, "f32")
                        for c in bank.chans:  # the float32 channelizers' state moves along (history, position)
                            c._advance(x32, n)
                    else:
                        zs = bank.process(raw)
                    for t, z in zip(members, zs):
                        t.after_block(z, tracker)
                if raw16 is not None:
                    self.integer_blocks += 0 if lo_needed else 1
                    self.split_blocks += 1 if lo_needed else 0
                for bi, (_, members) in enumerate(banks):
                    for ti, t in enumerate(members):  # (for finish(): which kernel produced this target's last block)
                        t.chan16_kernel = banks16[bi][0].chans[ti]._kernel.last_kernel if raw16 is not None else None
                _check_cancel("encode")
                done = hi

            tracker.status("flush outputs")
            for t in targets:
                t.finish()
            tracker.status("Processing complete")
            return [ProcessingResult(rate_probe, center_freq, t.target_freq, t.freq_offset, t.decimation, t.fs_channel,
                                     t.mix_sign, t.peak) for t in targets]
        except ProcessingCancelled:
            if not cfg.probe_only:
                paths = [t.output_path for t in targets] or [c.output_path for c in self.configs if c.output_path]
                for pth in paths:
                    with contextlib.suppress(OSError):
                        pth.unlink(missing_ok=True)
            raise
        finally:
            if "stager" in locals():
                stager.close()
            tracker.close()
