#!/usr/bin/env python3
"""Float32 captures: the float32 VALU kernel against the int16 planes on the matrix cores (round 3).

  python profiles/bench_f32.py [frames]     (default 300 M frames = 2.4 GB of cf32 at 10 MS/s, D = 104, 6401 taps)

Prints ms per capture and GS/s for: (a) k_channelize_v1 on the cf32 capture; (b) iqa_f32_split_s16 + ONE pass of the ring
kernel (a capture on the 2^-15 grid: low plane empty); (c) split + TWO passes + the z = z(hi) + 2^-15 z(lo) add (a
fractional capture), and the RMS difference of (c) against (a)."""
import sys
import time
from ctypes import c_int32, c_int64
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import iq_to_audio_amd as A  # noqa: E402
from iq_to_audio_amd import _dev as D  # noqa: E402
from iq_to_audio_amd import _native as N  # noqa: E402
from iq_to_audio_amd import dsp_plan as P  # noqa: E402
from iq_to_audio_amd.benchmark import synthetic_iq_s16  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
fs, f_off = 10e6, 25e3
d, _ = P.choose_decimation(fs, 96_000.0)
taps = A.design_channel_filter(fs, 12_500.0, d)
uniq = synthetic_iq_s16(fs, 2.0, f_off).reshape(-1)
rng = np.random.default_rng(3)
frac = (uniq.astype(np.float64) / 32768.0 * 0.9 + rng.normal(scale=1e-5, size=uniq.size)).astype(np.float32)
tile = torch.from_numpy(frac).to(D.device())
x = tile.repeat(-(-2 * n // tile.numel()))[: 2 * n].contiguous()
del tile
torch.cuda.synchronize()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def valu():
    return A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="f32").process(x, last_block=True)


hi, lo, flag = D.empty(2 * n, "int16"), D.empty(2 * n, "int16"), D.zeros(1, "int32")


def planes(two: bool):
    flag.zero_()
    N.call("iqa_f32_split_s16", N.ptr(x), c_int64(2 * n), c_int32(0), N.ptr(hi), N.ptr(lo), N.ptr(flag), N.stream_ptr())
    z = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="s16").process(hi, last_block=True)
    if two:
        zl = A.Channelizer(taps, sample_rate=fs, freq_offset=f_off, mix_sign=1, decimation=d, fmt="s16").process(lo, last_block=True)
        z.add_(zl, alpha=2.0 ** -15)
    return z


t_v, z_v = timed(valu, reps=2)
t_1, _ = timed(lambda: planes(False))
t_2, z_2 = timed(lambda: planes(True))
err = float((z_2 - z_v).abs().pow(2).mean().sqrt().item())
print(f"{n} frames of cf32 ({8 * n / 1e9:.1f} GB), D={d}, {len(taps)} taps; flag bits of the split: {int(flag.item())}")
print(f"(a) float32 VALU kernel            : {t_v * 1e3:8.2f} ms  {n / t_v / 1e9:7.1f} GS/s")
print(f"(b) split + one int16 ring pass    : {t_1 * 1e3:8.2f} ms  {n / t_1 / 1e9:7.1f} GS/s")
print(f"(c) split + two passes + add       : {t_2 * 1e3:8.2f} ms  {n / t_2 / 1e9:7.1f} GS/s   rms |z(c) - z(a)| = {err:.2e}")
