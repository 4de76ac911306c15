"""Host-side planning for the hot path (float64 / exact-integer scalar logic, no device work).

Mirrors the scalar rules of the reference's ``processing.py`` (chunk tuning, decimation
choice, Kaiser channel filter) and adds what the fused GPU channelizer needs: complex
taps pre-rotated by the NCO, the 64-bit fixed-point output rotation, and the folding
of ``iq_order`` into taps/flags so that the kernel's inner loop is a plain dot product.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass
from fractions import Fraction

import numpy as np

MAX_CHUNK = 4_194_304
TWO64 = 1 << 64
IQ_ORDERS = ("iq", "qi", "iq_inv", "qi_inv")
INGEST_SCALE = {"s16": 1.0 / 32768.0, "u8": 1.0 / 128.0, "f32": 1.0}
FMT_CODE = {"s16": 0, "u8": 1, "f32": 2}
FMT_NUMPY = {"s16": np.int16, "u8": np.uint8, "f32": np.float32}


#: (minimum sample rate, seconds of signal a chunk should hold), highest rate first -- reference processing.py:69-72
_CHUNK_SECONDS = ((5_000_000.0, 0.50), (2_000_000.0, 0.40), (0.0, 0.25))


def tune_chunk_size(sample_rate: float, requested: int) -> int:
    """Effective chunk length (reference processing.py:65-81): the requested one unless the rate calls for more --
    then the next power of two holding 0.25 s of signal (0.40 s from 2 MS/s, 0.50 s from 5 MS/s), never below the
    request and never above 4 Mi frames.

    The result is *semantic*, not a memory knob here: it fixes where the SSB AGC gain
    restarts and which prefix ``choose_mix_sign`` inspects.
    """
    floor = max(1, requested)
    if sample_rate <= 0:
        return floor
    seconds = next(s for rate, s in _CHUNK_SECONDS if sample_rate >= rate)
    wanted = int(round(sample_rate * seconds))
    if wanted <= floor:
        return floor
    wanted = min(wanted, MAX_CHUNK)
    return int(min(max(1 << (wanted - 1).bit_length(), floor), MAX_CHUNK))


def choose_decimation(sample_rate: float, fs_ch_target: float) -> tuple[int, float]:
    """(D, fs_channel) (reference processing.py:885-890; Python round = half-to-even)."""
    decimation = max(1, int(round(sample_rate / fs_ch_target)))
    fs_channel = sample_rate / decimation
    if fs_channel > fs_ch_target * 1.5:
        decimation = max(int(math.floor(sample_rate / fs_ch_target)), 1)
        fs_channel = sample_rate / decimation
    return decimation, fs_channel


def kaiser_beta(atten_db: float) -> float:
    """Kaiser's empirical beta(A) formula (what scipy.signal.kaiser_beta evaluates)."""
    a = abs(atten_db)
    if a > 50:
        return 0.1102 * (a - 8.7)
    if a > 21:
        return 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21)
    return 0.0


def design_channel_filter(sample_rate: float, bandwidth: float, decimation: int) -> np.ndarray:
    """Kaiser-windowed low-pass, float64, unity DC gain (reference processing.py:599-620).

    Same design rule as the reference (taps = clip(4*fs/max(1000, bw/2), 1024, 32768) made
    odd; cutoff = min(0.525*bw, 0.9*fs/(2D)); 80 dB Kaiser), evaluated directly with NumPy:
    h[n] = 2fc/fs * sinc(2fc/fs * (n - (N-1)/2)) * kaiser(N, beta), normalised to sum 1 --
    the windowed-sinc construction scipy.signal.firwin documents for a single low-pass band.
    """
    guard = max(1_000.0, bandwidth * 0.5)
    cutoff = min(bandwidth * 0.5 * 1.05, (sample_rate / (2.0 * max(decimation, 1))) * 0.9)
    if cutoff <= 0:
        raise ValueError("Invalid cutoff frequency for channel filter.")
    width = guard / sample_rate
    num_taps = int(np.clip(4.0 / max(width, 1e-8), 1024, 32768))
    if num_taps % 2 == 0:
        num_taps += 1
    beta = kaiser_beta(80.0)
    fc = cutoff / (0.5 * sample_rate)  # relative to Nyquist
    m = np.arange(num_taps, dtype=np.float64) - (num_taps - 1) / 2.0
    h = fc * np.sinc(fc * m) * np.kaiser(num_taps, beta)
    return np.asarray(h / h.sum(), dtype=np.float64)


def freq_ratio_turns(freq_offset: float, sample_rate: float, sign: int) -> int:
    """NCO phase advance per input sample as a 64-bit fraction of a turn.

    The reference mixes with exp(j*sign*inc*n), inc = -2*pi*f_off/fs (processing.py:287,293),
    i.e. -sign*f_off/fs turns per sample.  Floats are exact rationals, so the ratio is
    formed exactly and rounded once to 2^-64 turn.
    """
    theta = -sign * Fraction(freq_offset) / Fraction(sample_rate)
    theta -= math.floor(theta)
    return int(round(theta * TWO64)) % TWO64


@dataclass
class ChannelPlan:
    """Everything the fused channelizer kernel needs for one channel."""

    fmt: str
    ntaps: int
    decimation: int
    taps_window: np.ndarray  # complex64 [Lpad], window order, ingest scale folded in
    conj_sum: int
    rotate: int
    rot_step: int
    rot_base: int
    out_scale: complex
    taps_natural: np.ndarray | None = None  # complex128 [L], natural order g[k] (same folding), for the MFMA planner


def plan_channel(
    taps: np.ndarray,
    *,
    sample_rate: float,
    freq_offset: float,
    mix_sign: int,
    decimation: int,
    fmt: str = "s16",
    iq_order: str = "iq",
    padded_len: int | None = None,
) -> ChannelPlan:
    """Fold NCO + iq_order + ingest scale into complex taps and output-rotation constants.

    mix -> filter -> decimate of the reference equals
        z[m] = e^{-j 2 pi theta m D} * sum_k (h[k] e^{+j 2 pi theta k}) x[mD - k],
    theta = -sign*f_off/fs turns/sample ... with the sign convention below:
    the mixer multiplies sample n by e^{+j 2 pi W n / 2^64}; pulling e^{+j 2 pi W (mD)}
    out of the sum leaves taps g[k] = h[k] e^{-j 2 pi W k / 2^64}.
    """
    if iq_order not in IQ_ORDERS:
        raise ValueError(f"Unsupported iq_order '{iq_order}'")
    if fmt not in FMT_CODE:
        raise ValueError(f"unsupported sample format {fmt!r}")
    h = np.asarray(taps, dtype=np.float64)
    ntaps = h.size
    w = freq_ratio_turns(freq_offset, sample_rate, mix_sign)
    k = np.arange(ntaps, dtype=np.uint64)
    with np.errstate(over="ignore"):
        ph = np.uint64(w) * k  # wraps mod 2^64 == mod one turn
    turns = (ph >> np.uint64(11)).astype(np.float64) * (2.0**-53)
    g = h * np.exp(-2j * np.pi * turns)
    # raw frame r = a + jb (a = first value of the pair).  x = c*r or c*conj(r):
    #   iq: r | qi: j*conj(r) | iq_inv: conj(r) | qi_inv: -j*r      (IQReader._extract_iq)
    conj = iq_order in ("qi", "iq_inv")
    c = {"iq": 1.0 + 0j, "qi": 1j, "iq_inv": 1.0 + 0j, "qi_inv": -1j}[iq_order]
    if conj:
        g = np.conj(g)  # sum g*conj(r) = conj(sum conj(g)*r)
    g = g * INGEST_SCALE[fmt]
    lpad = padded_len if padded_len is not None else -(-ntaps // 256) * 256
    win = np.zeros(lpad, dtype=np.complex64)
    win[:ntaps] = g[::-1].astype(np.complex64)  # window order: win[i] multiplies x[n0-(L-1)+i]
    return ChannelPlan(
        fmt=fmt, ntaps=ntaps, decimation=int(decimation), taps_window=win, conj_sum=int(conj), rotate=1,
        rot_step=(w * int(decimation)) % TWO64, rot_base=0, out_scale=c, taps_natural=g.astype(np.complex128),
    )


MFMA_Q = 64  # q slots (tap rows) per pass and output component in the int8-MFMA kernel
MFMA_KSTEP_BYTES = 8192  # tap fragments per k step: 4 row tiles x 2 pieces x 64 lanes x 16 B
MFMA_MAX_KSTEPS_PER_PASS = 16  # 128 KiB of fragments leaves room for >= 1888 outputs of accumulators in LDS


@dataclass
class MfmaGroup:
    """Tap rows 64*q+1 .. 64*q+64 of the filter (or, ``residual``, what an earlier group left of them), quantised on
    their own scale."""

    afrag: np.ndarray  # int8 [ksteps, 4, 2, 64, 16] tap fragments in MFMA lane order
    unit: float  # value of one tap LSB
    tq: np.ndarray  # int32 [128, Kpad] quantised taps T = 256*q1 + q2
    q: int = 0  # tap-row group (the data rows a lane of this group streams start 64*q rows earlier)
    residual: bool = False  # the taps of this group are the quantisation residue of the group in front of it
    high_only: bool = False  # every low tap byte q2 is zero: the ring kernels skip the q2*hi product (two MFMAs per k step)


@dataclass
class MfmaPass:
    group: int  # index into MfmaPlan.groups
    k_first: int
    k_count: int
    c_re: float  # 128 * sum(T) over the real-output rows and this pass's k range (low-byte bias)
    c_im: float


@dataclass
class MfmaPlan:
    """Host-side operands of ``iqa_channelize_mfma`` (see csrc/channelize_mfma.hip)."""

    ksteps: int
    groups: list
    passes: list
    err_norm: float = 0.0  # sqrt(sum_k |T_k u - g_k|^2) per unit of full scale: the z error for a white input of unit RMS
    #: RMS of the part of the z error that does NOT scale with the input: the q2*lo' products the int16 kernels drop
    #: (low tap byte x low data byte), for data whose low bytes are uniform (any capture well above 8 bits)
    floor_rms: float = 0.0

    # single-pass conveniences (tests, and the common ceil(L/D) <= 64, D <= 256 case)
    @property
    def afrag(self):
        return self.groups[0].afrag

    @property
    def unit(self):
        return self.groups[0].unit

    @property
    def tq(self):
        return self.groups[0].tq

    @property
    def c_re(self):
        return sum(p.c_re for p in self.passes if p.group == 0)

    @property
    def c_im(self):
        return sum(p.c_im for p in self.passes if p.group == 0)

    def z_error_rms(self, wideband_rms: float) -> float:
        """Expected RMS error of z for a capture of the given wideband RMS (fraction of full scale)."""
        return math.hypot(self.err_norm * wideband_rms, self.floor_rms)


def mfma_supported(plan: ChannelPlan) -> bool:
    """int16 captures (every matrix-core kernel) and uint8 captures (the row-staged ring kernel only)."""
    return plan.fmt in ("s16", "u8") and plan.taps_natural is not None


@functools.lru_cache(maxsize=32)
def _mfma_layout(ntaps: int, decimation: int, group: int):
    """Index tables of the tap-fragment layout; they depend only on (L, D, q-group)."""
    L, D = ntaps, decimation
    ksteps = -(-2 * D // 32)
    q = np.arange(1, MFMA_Q + 1, dtype=np.int64)[:, None] + MFMA_Q * group
    rho = np.arange(D, dtype=np.int64)[None, :]
    k = q * D - 1 - rho
    ok = (k >= 0) & (k < L)
    kc = np.clip(k, 0, L - 1)
    lane = np.arange(64)
    rows = (np.arange(4) * 32)[:, None] + (lane & 31)[None, :]  # [rt, lane]
    cols = (32 * np.arange(ksteps))[:, None, None] + (16 * (lane >> 5))[None, :, None] + np.arange(16)[None, None, :]
    flat = (rows[None, :, :, None] * (32 * ksteps) + cols[:, None, :, :]).astype(np.int64)  # [ks, rt, lane, j]
    return ksteps, kc, ok, flat


_LOW_BYTE_VAR = (256.0 * 256.0 - 1.0) / 12.0  # variance of a uniform low data byte lo' in [-128, 127]


def _quantise_rows(a: np.ndarray, acc32: bool, high_byte_only: bool, col_ranges=None):
    """(unit, T, q1, q2) for one 128-row tap matrix: T = rint(a / unit) = 256*q1 + q2 with both bytes signed,
    unit = max|a| / 32639 -- enlarged, with ``acc32``, until 256*S1 + S2 of a component cannot overflow an int32
    for ANY int16 input (``col_ranges``: the column ranges that are summed into one int32 -- a pass of the kernel covers
    one k-step range and hands its sum on in float64).  ``high_byte_only``: q2 = 0 (T a multiple of 256): the int16
    kernels then drop nothing -- their q2*lo' term is identically zero -- at the price of taps of ~6 bits."""
    amax = float(np.abs(a).max())
    unit = amax / 32639.0 if amax > 0 else 1.0
    while True:
        if high_byte_only:
            q1 = np.clip(np.rint(a * (1.0 / (256.0 * unit))), -127, 127).astype(np.int32)
            q2 = np.zeros_like(q1)
            t = q1 << 8
        else:
            t = np.rint(a * (1.0 / unit)).astype(np.int32)
            q2 = ((t + 128) & 255) - 128
            q1 = (t - q2) >> 8
        if not acc32:
            break
        # |256*S1 + S2| <= sum 128*(257|q1| + |q2|) over one component's rows (|hi|, |lo'| <= 128)
        w = 128 * (257 * np.abs(q1).astype(np.int64) + np.abs(q2))
        bound = max(int(w[r, c0:c1].sum()) for r in (slice(0, MFMA_Q), slice(MFMA_Q, 2 * MFMA_Q))
                    for c0, c1 in (col_ranges or [(0, a.shape[1])]))
        if bound < 2**31 - 1:
            break
        unit *= max(1.02, bound / (2**31 - 1) * 1.001)
    return unit, t, q1, q2


def plan_mfma(plan: ChannelPlan, acc32: bool = False, max_ksteps: int | None = None, residual: bool = False) -> MfmaPlan:
    """Quantise the (already NCO-rotated, scaled) taps to 16-bit fixed point and lay them out as
    the A operand of v_mfma_i32_32x32x32_i8.

    ``acc32``: the ring kernel keeps 256*S1 + S2 of an output component in ONE int32.  The tap unit is then
    enlarged (taps of ~14 bits for the standard 12.5 kHz filters) until
    sum_k 128*(257|q1_k| + |q2_k|) < 2^31 over the rows of a component, so that the sum cannot overflow for
    ANY int16 input; everything stays exact integer arithmetic.

    ``residual`` (the "fine" precision of the pipeline): every tap-row group becomes TWO groups that stream the same data
    rows -- the taps themselves and what their quantisation left over, quantised again on a unit of its own (~1/20 of
    the first one's LSB under the int32 bound) -- whose partial sums ``iqa_mfma_combine`` adds.  For int16 captures the
    first of the two keeps the high tap byte only: the kernels drop the (low tap byte) x (low data byte) products, an
    error of the same size as the tap rounding itself, and with q2 = 0 there is nothing to drop; its result is then the
    EXACT product of its (coarse) taps and the residual group carries everything else.  Twice the matrix work, ~20x
    less error, same kernels.

    Rows: row = comp*64 + (q-1) within a q-group of 64 tap rows; columns kap = 2*rho + c over one
    data row of D frames:
        A[(re,q)][2rho] = Re g[qD-1-rho]   A[(re,q)][2rho+1] = -Im g[qD-1-rho]
        A[(im,q)][2rho] = Im g[qD-1-rho]   A[(im,q)][2rho+1] =  Re g[qD-1-rho]
    T = rint(A/u), u = max|A|/32639 per group; T = 256*q1 + q2 with both bytes signed.
    Fragment order (verified on hardware): lane l holds row l&31, k = 16*(l>>5) + j.
    Filters with ceil(L/D) > 64 get several q-groups, decimations whose fragments exceed LDS get
    several k-step ranges; every (group, range) is one pass of the kernel.
    """
    if not mfma_supported(plan):
        raise ValueError("MFMA channelizer needs an int16 or uint8 capture")
    g = plan.taps_natural
    D = plan.decimation
    n_groups = max(1, -(-(-(-plan.ntaps // D)) // MFMA_Q))
    groups, passes = [], []
    err_sq = floor_sq = 0.0
    ksteps = -(-2 * D // 32)
    n_chunks = -(-ksteps // (max_ksteps or MFMA_MAX_KSTEPS_PER_PASS))  # k-step ranges: one pass each
    bounds = [round(i * ksteps / n_chunks) for i in range(n_chunks + 1)]
    s16 = plan.fmt == "s16"
    for gi in range(n_groups):
        _, kc, ok, flat = _mfma_layout(plan.ntaps, D, gi)
        kpad = 32 * ksteps
        gk = g[kc]
        gre = np.where(ok, gk.real, 0.0)
        gim = np.where(ok, gk.imag, 0.0)
        a = np.zeros((2 * MFMA_Q, kpad), dtype=np.float64)
        a[:MFMA_Q, 0 : 2 * D : 2] = gre
        a[:MFMA_Q, 1 : 2 * D : 2] = -gim
        a[MFMA_Q:, 0 : 2 * D : 2] = gim
        a[MFMA_Q:, 1 : 2 * D : 2] = gre
        left = a
        for part in range(2 if residual else 1):
            # (uint8 data are one piece: the kernels' T*v is exact whatever the low tap byte holds)
            unit, t, q1, q2 = _quantise_rows(left, acc32, high_byte_only=residual and part == 0 and s16,
                                             col_ranges=[(32 * bounds[i], 32 * bounds[i + 1]) for i in range(n_chunks)])
            left = left - t * unit
            if s16:
                floor_sq += unit * unit * _LOW_BYTE_VAR * float((q2.astype(np.float64) ** 2).sum())
            frag = np.empty((ksteps, 4, 2, 64, 16), dtype=np.int8)
            frag[:, :, 0] = q1.reshape(-1)[flat]
            frag[:, :, 1] = q2.reshape(-1)[flat]
            groups.append(MfmaGroup(frag, unit, t, q=gi, residual=part == 1, high_only=bool(s16 and not np.any(q2))))
            for ci in range(n_chunks):
                k0, k1 = bounds[ci], bounds[ci + 1]
                sl = t[:, 32 * k0 : 32 * k1]
                # int16 data are split v = 256*hi + lo' + 128: the 128 makes a constant 128*sum(T); uint8 data have one piece
                bias = 128.0 if s16 else 0.0
                passes.append(MfmaPass(len(groups) - 1, k0, k1 - k0, bias * float(sl[:MFMA_Q].sum(dtype=np.int64)),
                                       bias * float(sl[MFMA_Q:].sum(dtype=np.int64))))
        err_sq += 0.5 * float((left**2).sum())  # every complex tap sits in the matrix twice (re and im rows)
    return MfmaPlan(ksteps, groups, passes, math.sqrt(err_sq) / INGEST_SCALE[plan.fmt], math.sqrt(floor_sq))


def mfma_interior(consumed: int, n_frames: int, m_first: int, n_out: int, decimation: int, ksteps: int,
                  n_groups: int = 1):
    """(m_a, m_b): the sub-range of outputs [m_first, m_first+n_out) whose whole MFMA read range
    (columns m-64*n_groups .. m+29, each 16*ksteps frames from frame b*D+1) lies inside this block's frames."""
    d = decimation
    m_a = max(m_first, MFMA_Q * n_groups + -(-(consumed - 1) // d))  # consumed < 0: a lead-in in front of frame 0
    m_b = min(m_first + n_out, (n_frames + consumed - 16 * ksteps) // d - 30)
    return (m_a, m_b) if m_b > m_a else (m_first, m_first)


def plan_plain_fir(taps: np.ndarray, padded_len: int | None = None) -> ChannelPlan:
    """Real taps, complex64 in/out, no decimation, no rotation (the OverlapSaveFIR stage)."""
    h = np.asarray(taps, dtype=np.float64)
    lpad = padded_len if padded_len is not None else -(-h.size // 256) * 256
    win = np.zeros(lpad, dtype=np.complex64)
    win[: h.size] = h[::-1].astype(np.complex64)
    return ChannelPlan("f32", h.size, 1, win, 0, 0, 0, 0, 1.0 + 0j)


def chunk_output_starts(chunk: int, decimation: int, first_frame: int, n_frames: int) -> np.ndarray:
    """Indices (relative to the first output of the block) of the first decimated sample of
    every reference chunk that starts inside frames [first_frame, first_frame+n_frames).

    ``first_frame`` must be a multiple of ``chunk``.  These are the points where the SSB AGC
    gain restarts (decoders/ssb.py:72) and where per-chunk statistics are cut.
    """
    if first_frame % chunk:
        raise ValueError("blocks must start on a chunk boundary")
    d = decimation
    m_first = -(-first_frame // d)
    starts = np.arange(first_frame, first_frame + n_frames, chunk, dtype=np.int64)
    return (-(-starts // d) - m_first).astype(np.int64)


# ---- 48 kHz resampler plan (build-defined spec; see DESIGN.md "48 kHz stage") -----------------

RS_ZERO_CROSSINGS = 16
RS_CUTOFF = 0.97
RS_KAISER_BETA = 9.0
RS_OUT_RATE = 48_000


@dataclass
class ResamplerPlan:
    in_rate: int
    up: int
    down: int
    half_taps: int  # T: table rows hold taps t = -T..T
    table: np.ndarray  # float64 [up, 2T+1]

    def n_out(self, n_in: int) -> int:
        return -(-n_in * self.up // self.down)


def plan_resampler(fs_channel: float, out_rate: int = RS_OUT_RATE) -> ResamplerPlan:
    """Polyphase table of the zero-phase Kaiser-sinc prototype (planned once per declared input rate: the table of
    the 96 154 -> 48 000 Hz case has 1.6 M entries and its Bessel window costs ~55 ms of host NumPy -- most of a file ->
    WAV run on the reference's own 5 s benchmark capture before it was cached).  The plan and its table are shared: do
    not modify them.

    The declared input rate is round(fs_channel), exactly what the reference tells ffmpeg
    (processing.py:389-397).  Prototype (common rate up*in_rate): h[i] = sinc(0.97*i/M) *
    kaiser(beta=9) over |i| <= 16*M, M = max(up, down), scaled to sum(h) = up.
    Row p of the table holds h[p + t*up] for t = -T..T (zero outside the support).
    """
    return _plan_resampler(max(1, int(round(fs_channel))), int(out_rate))


@functools.lru_cache(maxsize=16)
def _plan_resampler(rin: int, out_rate: int) -> ResamplerPlan:
    g = math.gcd(out_rate, rin)
    up, down = out_rate // g, rin // g
    m = max(up, down)
    half = RS_ZERO_CROSSINGS * m
    i = np.arange(-half, half + 1, dtype=np.float64)
    u = i / half
    win = np.i0(RS_KAISER_BETA * np.sqrt(np.clip(1.0 - u * u, 0.0, 1.0))) / np.i0(RS_KAISER_BETA)
    h = np.sinc(RS_CUTOFF * i / m) * win
    h *= up / h.sum()
    t_half = -(-half // up)
    idx = np.arange(up, dtype=np.int64)[:, None] + np.arange(-t_half, t_half + 1, dtype=np.int64)[None, :] * up
    ok = np.abs(idx) <= half
    table = np.ascontiguousarray(np.where(ok, h[np.clip(idx + half, 0, 2 * half)], 0.0), dtype=np.float64)
    table.setflags(write=False)
    return ResamplerPlan(rin, up, down, int(t_half), table)
